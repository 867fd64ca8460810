#!/bin/bash
# When each wave of the stereo fast path ran and where (diagnostic build -DVPZ_WAVE_TIMES: two clock reads per wave).
# usage: bash tools/wave_times.sh <which: real|ola> [VPZ_DUAL_EXP]      (run through gpurun)
set -e
W=${1:-real}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
export VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_WAVE_TIMES
python -c "import __graft_entry__ as g; g.build()"
for e in ${2:-0}; do
VPZ_DUAL_EXP=$e VPZ_STAMPS_DUMP=$GRAFT_REPO_ROOT/gpurun_out/r3/wave_times_${W}_exp$e.csv python tools/kbench_synth.py --steps 3 --which $W 2>&1 | grep "configs" | tail -2
done
python tools/experiments/wave_times_fit.py gpurun_out/r3/wave_times_${W}_exp0.csv
unset VPZ_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()"
