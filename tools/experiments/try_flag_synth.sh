#!/bin/bash
# Tuning: the fused workloads (configs[3] 6 ch, configs[4] share, configs[2], all-long line) with an extra compile flag against the
# product build, alternating on one box; the flagged build first runs the transform / stereo / group parity tests.
# usage: bash tools/try_flag_synth.sh -DVPZ_SOMETHING [out.txt]
F=$1
cd "$GRAFT_REPO_ROOT"
OUT=${2:-gpurun_out/try_flag.txt}
mkdir -p $(dirname $OUT)
run() { for w in floor real ola olalong; do python tools/kbench_synth.py --which $w --steps 40 2>&1 | tail -1 || return 1; done; }
{
echo "== product"; run || exit 1
VPZ_EXTRA_HIPCC_FLAGS=$F python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
export VPZ_EXTRA_HIPCC_FLAGS=$F
python -m pytest tests/test_imdct_gpu.py tests/test_dual_gpu.py tests/test_synth_gpu.py -x -q 2>&1 | tail -2 || exit 1
echo "== $F"; run || exit 1
echo "== $F"; run || exit 1
unset VPZ_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
echo "== product"; run || exit 1
} 2>&1 | tee $OUT
