#!/bin/bash
# Kernel-level comparison (rocprofv3 --kernel-trace --stats) of the stereo fast path's tuning variants on one box.
# usage: bash tools/prof_exp.sh "<list of VPZ_DUAL_EXP values>" [which] [outdir]
set -e
LIST=${1:-"0 1 2 3"}
W=${2:-real}
OUT=${3:-gpurun_out/r3/prof_exp}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for rep in 1 2; do
for e in $LIST; do
  export VPZ_DUAL_EXP=$e
  rocprofv3 --kernel-trace --stats -d "$OUT/${W}_exp${e}_$rep" -o k --output-format csv -- python tools/kbench_synth.py --steps 10 --which $W > "$OUT/${W}_exp${e}_$rep.log" 2>&1
  echo "== $W VPZ_DUAL_EXP=$e (rep $rep)  $(grep -h configs "$OUT/${W}_exp${e}_$rep.log" | head -1)"
  python tools/prof_summary.py "$OUT/${W}_exp${e}_$rep/k_kernel_stats.csv" | grep -v unwrap
done
done
