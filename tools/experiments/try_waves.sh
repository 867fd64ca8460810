#!/bin/bash
# Tuning: the stereo fast path with another workgroup size (-DVPZ_DUAL_WAVES=N) against the product build, same box.
# usage: bash tools/try_waves.sh <N>
N=${1:-10}
cd "$GRAFT_REPO_ROOT"
for w in real ola olalong; do echo "product  : $(python tools/kbench_synth.py --which $w --steps 40 2>&1 | grep -v amdgpu | tail -1)"; done
VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_DUAL_WAVES=$N python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
export VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_DUAL_WAVES=$N
python -m pytest tests/test_dual_gpu.py -q -x 2>&1 | tail -2
for w in real ola olalong; do echo "waves $N : $(python tools/kbench_synth.py --which $w --steps 40 2>&1 | grep -v amdgpu | tail -1)"; done
unset VPZ_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
for w in real; do echo "product  : $(python tools/kbench_synth.py --which $w --steps 40 2>&1 | grep -v amdgpu | tail -1)"; done
