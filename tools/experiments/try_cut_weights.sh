#!/bin/bash
# Tuning: the host's run-cutting weights (cost of a short block alone / at the head of a batch, of a block riding in a batch, in
# eighths of a long pass) on the workloads with short blocks, alternating on one box.   usage: tools/try_cut_weights.sh [out.txt]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/try_cut_weights.txt}
mkdir -p $(dirname $OUT)
{
for i in 1 2; do
  for w in ${WEIGHTS:-6,3 8,2 8,3 7,2}; do
    echo "== VPZ_CUT_WEIGHTS=$w"
    VPZ_CUT_WEIGHTS=$w python tools/kbench_synth.py --which real --steps 40 2>&1 | tail -1 || exit 1
    VPZ_CUT_WEIGHTS=$w python tools/kbench_synth.py --which ola --steps 40 2>&1 | tail -1 || exit 1
  done
done
} 2>&1 | tee $OUT
