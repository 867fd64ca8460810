#!/bin/bash
# Timing experiment (wrong results): configs[3] with the interleaved packet landed in the group's LDS rows by LDS-DMA
# (VPZ_SYNTH_ABLATE=2048) against the product's register prefetch + ds_write de-interleave, alternating on one box;
# also without the de-interleave altogether (32) as the bound.   usage: bash tools/try_lds_dma.sh [out.txt]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/try_lds_dma.txt}
mkdir -p $(dirname $OUT)
{
for i in 1 2 3; do
  for ab in ${ABS:-0 2048 32}; do
    echo "== VPZ_SYNTH_ABLATE=$ab"
    VPZ_SYNTH_ABLATE=$ab python tools/kbench_synth.py --which floor --steps 40 2>&1 | tail -1 || exit 1
  done
done
} 2>&1 | tee $OUT
