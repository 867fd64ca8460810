# kernel time alone (rocprofv3 --kernel-trace --stats) of builds under vorbispizza_amd/lib_ab/<name>/ against the product build, alternating:
#   bash tools/experiments/ab_builds_kernel_time.sh out.txt workload kernel_substring name1 name2 ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$1; W=$2; K=$3; shift; shift; shift
: > "$OUT"
for round in 1 2 3; do
  for which in "$@" product; do
    if [ "$which" = product ]; then unset VPZ_LIB_DIR; else export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/$which"; fi
    D=gpurun_out/abk_tmp; rm -rf $D
    rocprofv3 --kernel-trace --stats -d $D -o k --output-format csv -- python tools/kbench_synth.py --which $W --steps 60 > /dev/null 2>&1
    line=$(python tools/prof_summary.py $D/k_kernel_stats.csv | grep "$K" | head -1)
    echo "round $round  $which  $line" >> "$OUT"
  done
done
rm -rf gpurun_out/abk_tmp
