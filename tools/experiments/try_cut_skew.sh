#!/bin/bash
# Tuning: the skew of the run cutting (per mille of a run's cost moved from the grid's second half to its first; for runs of equal
# length: any value > 0 = one frame) on the stereo workloads, alternating on one box.   usage: tools/try_cut_skew.sh [out.txt]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/try_cut_skew.txt}
mkdir -p $(dirname $OUT)
{
for i in 1 2; do
  for w in ${SKEWS:-0 25}; do
    echo "== VPZ_CUT_SKEW=$w"
    for which in real ola olalong; do VPZ_CUT_SKEW=$w python tools/kbench_synth.py --which $which --steps 40 2>&1 | tail -1 || exit 1; done
  done
done
} 2>&1 | tee $OUT
