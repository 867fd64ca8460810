#!/bin/bash
# A/B on one box: group mode with the interleaved packet landed in LDS as it is (LDS-DMA, coupling at pick-up) against the
# register prefetch + de-interleaving stores + coupling pass (VPZ_GROUP_DMA=0 (product)), alternating.  usage: tools/ab_group_dma.sh [out.txt]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/ab_group_dma.txt}
mkdir -p $(dirname $OUT)
{
for i in 1 2 3; do
  for nd in 1 0; do
    echo "== VPZ_GROUP_DMA=$nd"
    VPZ_GROUP_DMA=$nd python tools/kbench_synth.py --which floor --steps 40 2>&1 | tail -1 || exit 1
  done
done
for nd in 1 0; do
  echo "== VPZ_GROUP_DMA=$nd"
  VPZ_GROUP_DMA=$nd python tools/experiments/kbench_layouts.py 2>&1 | tail -6 || exit 1
  VPZ_NO_DUAL=1 VPZ_GROUP_DMA=$nd python tools/kbench_synth.py --which real --steps 40 2>&1 | tail -1 || exit 1
done
} 2>&1 | tee $OUT
