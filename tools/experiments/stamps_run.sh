#!/bin/bash
# Phase stamps of the fused kernels (diagnostic build, -DVPZ_STAMPS): cycles per wave and phase, printed per call.
# usage: bash tools/stamps_run.sh <which: real|ola|floor>     (run through gpurun; rebuilds the library twice)
set -e
W=${1:-real}
cd "$GRAFT_REPO_ROOT"
VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_STAMPS python -c "import __graft_entry__ as g; g.build()"
mkdir -p gpurun_out/r3
VPZ_STAMPS_DUMP=$GRAFT_REPO_ROOT/gpurun_out/r3/stamps_waves_$W.csv VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_STAMPS python tools/kbench_synth.py --steps 3 --which $W 2>&1 | grep "stamps\|configs" | tail -6
python -c "import __graft_entry__ as g; g.build()"   # back to the product build
