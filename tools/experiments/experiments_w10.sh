#!/bin/bash
# 10-wave workgroups of the stereo fast path (-DVPZ_DUAL_WAVES=10, one per CU) against the product's 4-wave ones, alternating processes
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r5_w10.txt}
: > $OUT
for round in 1 2 3; do
  for which in w10 product; do
    if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
    for w in olalong real; do
      echo "round $round $which $w: $(python tools/ab_decoders.py --which $w --rounds 5 --variants default VPZ_DUAL_RUN=6 VPZ_DUAL_RUN=12 2>&1 | grep -v amdgpu | tr '\n' '|' | cut -c1-420)" >> $OUT
    done
  done
done
cat $OUT
