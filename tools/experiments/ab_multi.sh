#!/bin/bash
# Several builds (vorbispizza_amd/lib_ab/<name>, "product" = vorbispizza_amd/lib) on one box, alternating, one workload.
#   usage: bash tools/ab_multi.sh <out.txt> <workload> <rounds> <name> [<name> ...]
OUT=$1; W=$2; R=$3; shift; shift; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for round in $(seq 1 $R); do
  for which in "$@"; do
    if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
    line=$(python tools/kbench_synth.py --which $W --steps 40 2>&1 | grep -E 'configs|north_star' | tail -1)
    echo "round $round  $which  $line" | tee -a "$OUT"
  done
done
