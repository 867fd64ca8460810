#!/bin/bash
# Tuning: the stereo fast path as ONE workgroup of 8 waves per CU (both wave slots of a SIMD belong to waves of the same age)
# against the product's two workgroups of 4 (where the second to arrive runs slower and the run cutting leans against it).
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/try_dual_waves8.txt}
mkdir -p $(dirname $OUT)
run() { for w in real ola olalong; do python tools/kbench_synth.py --which $w --steps 40 2>&1 | tail -1 || return 1; done; }
{
echo "== product (2 x 4 waves, skew 25)"; run || exit 1
echo "== product, VPZ_CUT_SKEW=0"; VPZ_CUT_SKEW=0 run || exit 1
VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_DUAL_WAVES=8 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
export VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_DUAL_WAVES=8
python -m pytest tests/test_dual_gpu.py -x -q 2>&1 | tail -2 || exit 1
echo "== 1 x 8 waves, VPZ_CUT_SKEW=0"; VPZ_CUT_SKEW=0 run || exit 1
echo "== 1 x 8 waves, skew 25"; run || exit 1
echo "== 1 x 8 waves, VPZ_CUT_SKEW=0"; VPZ_CUT_SKEW=0 run || exit 1
unset VPZ_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
echo "== product (2 x 4 waves, skew 25)"; run || exit 1
} 2>&1 | tee $OUT
