#!/bin/bash
# A/B of two BUILDS on the slow-path workloads (tools/kbench_slow_paths.py), alternating on one GPU box, then the per-kernel times of
# each under rocprofv3.  usage: bash tools/ab_slow_paths.sh <name under vorbispizza_amd/lib_ab> [out.txt]
NAME=${1:?name}
OUT=${2:-gpurun_out/r4/ab_slow_$NAME.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for round in 1 2 3; do
  for which in "$NAME" product; do
    if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
    python tools/kbench_slow_paths.py 2>&1 | grep -E '^\((a|b)\)' | cut -c1-60,150- | sed "s/^/round $round  $which  /" | tee -a "$OUT"
  done
done
for which in "$NAME" product; do
  if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
  echo "== kernels, $which" | tee -a "$OUT"
  bash tools/prof_slow_paths.sh 2>&1 | grep -E "floor0|dual|unwrap" | tee -a "$OUT"
done
