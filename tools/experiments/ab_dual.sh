#!/bin/bash
# A/B on one box: the stereo fast path (synth_dual_kernel) against the routes it replaces (VPZ_NO_DUAL=1), alternating.
# Usage: tools/ab_dual.sh [rounds]      (writes to stdout)
R=${1:-2}
for i in $(seq $R); do
  for nd in 0 1; do
    echo "== VPZ_NO_DUAL=$nd (round $i)"
    VPZ_NO_DUAL=$nd python tools/kbench_synth.py --which real --steps 40 || exit 1
    VPZ_NO_DUAL=$nd python tools/kbench_synth.py --which ola --steps 40 || exit 1
    VPZ_NO_DUAL=$nd python tools/experiments/kbench_short_long.py || exit 1
  done
done
