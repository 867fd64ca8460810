#!/bin/bash
# Round 5: chained runs (kPreNeighbour) against recomputed predecessors, and the preferred run length, on one box.
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r5_chain_ab.txt}
: > $OUT
for rep in 1 2; do
for cfg in "VPZ_NO_CHAIN=1" "VPZ_DUAL_RUN=32" "VPZ_DUAL_RUN=16" "VPZ_DUAL_RUN=12" "VPZ_DUAL_RUN=8" "VPZ_DUAL_RUN=6" "VPZ_DUAL_RUN=4"; do
  for w in olalong ola real; do
    echo "$cfg $w: $(env $cfg python tools/kbench_synth.py --which $w --steps 40 2>&1 | tail -1)" >> $OUT
  done
done
done
cat $OUT
