# several builds under vorbispizza_amd/lib_ab/<name>/ against the product build, alternating: bash tools/experiments/ab_builds_many.sh out.txt "wl1 wl2" name1 name2 ...
cd "$GRAFT_REPO_ROOT"
OUT=$1; WL=$2; shift; shift
: > "$OUT"
for round in 1 2 3; do
  for w in $WL; do
    for which in "$@" product; do
      if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
      line=$(python tools/kbench_synth.py --which $w --steps 40 2>&1 | grep -E 'configs|north_star' | tail -1)
      echo "round $round  $which  $line" >> "$OUT"
    done
  done
done
