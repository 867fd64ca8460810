#!/bin/bash
# Everything profiles/r5_* of the measurement kind is made from, on the GPU box (each part fits one gpurun limit):
#   bash tools/r5_profiles.sh stamps     traffic stamps (PMC passes) + kernel stats / timelines / PMC traffic of the bench and the workloads
#   bash tools/r5_profiles.sh counters   the contract line alone under rocprofv3, SQ counters of the fused workloads, slow paths, the final bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
case "${1:-stamps}" in
stamps)
  python tools/stamp_traffic.py > gpurun_out/r5/stamp_traffic.log 2>&1; tail -4 gpurun_out/r5/stamp_traffic.log
  cp profiles/traffic_stamp.json gpurun_out/r5/traffic_stamp.json
  bash tools/collect_profiles.sh r5 > gpurun_out/r5/collect.log 2>&1; tail -30 gpurun_out/r5/collect.log
  ;;
counters)
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" &&
    rocprofv3 --kernel-trace --stats -d gpurun_out/r5/contract_line -o b --output-format csv -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r5/contract_line.json 2> gpurun_out/r5/contract_line.err
    python tools/prof_summary.py gpurun_out/r5/contract_line/b_kernel_stats.csv > gpurun_out/r5/contract_line_kernel_stats.txt; cat gpurun_out/r5/contract_line_kernel_stats.txt; tail -c 700 gpurun_out/r5/contract_line.json )
  KERNEL=synth_dual WHICH=olalong bash tools/pmc_sq_counters.sh gpurun_out/r5/pmc_SQ_north_star_line.txt > /dev/null 2>&1
  KERNEL=synth_dual WHICH=real bash tools/pmc_sq_counters.sh gpurun_out/r5/pmc_SQ_config4.txt > /dev/null 2>&1
  KERNEL=synth_kernel WHICH=floor bash tools/pmc_sq_counters.sh gpurun_out/r5/pmc_SQ_config3.txt > /dev/null 2>&1
  python tools/kbench_slow_paths.py > gpurun_out/r5/slow_paths.txt 2>&1; cat gpurun_out/r5/slow_paths.txt
  bash tools/prof_slow_paths.sh > gpurun_out/r5/slow_paths_kernels.txt 2>&1
  python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; tail -c 600 gpurun_out/r5/bench_final.json
  ;;
esac
ls gpurun_out/r5
