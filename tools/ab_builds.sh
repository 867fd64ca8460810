#!/bin/bash
# A/B of two BUILDS of the libraries on one GPU box, alternating: vorbispizza_amd/lib_ab/<name>/ (VPZ_LIB_DIR) against the
# product build in vorbispizza_amd/lib/.  usage: bash tools/ab_builds.sh <name> [out.txt] [workloads...]
#   (build the other one in a worktree of the commit in question and copy its two .so files there; lib_ab/ is git-ignored
#   but travels to the GPU box)
NAME=${1:-r3}
OUT=${2:-gpurun_out/r4/ab_builds_$NAME.txt}
shift; shift
WL=${@:-floor real ola olalong}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for round in 1 2 3; do
  for w in $WL; do
    for which in "$NAME" product; do
      if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
      line=$(python tools/kbench_synth.py --which $w --steps 40 2>&1 | grep -E 'configs|north_star' | tail -1)
      echo "round $round  $which  $line" | tee -a "$OUT"
    done
  done
done
