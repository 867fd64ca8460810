# A/B of builds of synth_big.hip under vorbispizza_amd/lib_ab/<name>/ against the product build, alternating, on the two big-block workloads
# of tools/kbench_slow_paths.py (c).   usage: bash tools/experiments/ab_big_variants.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
  for which in product "$@"; do
    if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
    python tools/kbench_slow_paths.py 2>&1 | grep "^(c) block sizes" | sed 's/(synth_big.*HBM)//' | while read line; do echo "round $round  $which  $line"; done
  done
done
