#!/bin/bash
# Everything profiles/r4_* is made from, on the GPU box, in two calls (each fits one gpurun limit):
#   bash tools/r4_profiles.sh stamps     traffic stamps (PMC passes) + kernel stats / timelines / PMC traffic of the bench and the workloads
#   bash tools/r4_profiles.sh counters   SQ counters of the four fused workloads, slow paths, the final bench lines
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
case "${1:-stamps}" in
stamps)
  python tools/stamp_traffic.py > gpurun_out/r4/stamp_traffic.log 2>&1; tail -4 gpurun_out/r4/stamp_traffic.log
  cp profiles/traffic_stamp.json gpurun_out/r4/traffic_stamp.json
  bash tools/collect_profiles.sh r4 > gpurun_out/r4/collect.log 2>&1; tail -30 gpurun_out/r4/collect.log
  ;;
counters)
  KERNEL=synth_dual WHICH=olalong bash tools/pmc_sq_counters.sh gpurun_out/r4/pmc_SQ_north_star_line.txt > /dev/null 2>&1
  KERNEL=synth_dual WHICH=ola bash tools/pmc_sq_counters.sh gpurun_out/r4/pmc_SQ_config2.txt > /dev/null 2>&1
  KERNEL=synth_dual WHICH=real bash tools/pmc_sq_counters.sh gpurun_out/r4/pmc_SQ_config4.txt > /dev/null 2>&1
  KERNEL=synth_kernel WHICH=floor bash tools/pmc_sq_counters.sh gpurun_out/r4/pmc_SQ_config3.txt > /dev/null 2>&1
  python tools/kbench_slow_paths.py > gpurun_out/r4/slow_paths.txt 2>&1; cat gpurun_out/r4/slow_paths.txt
  bash tools/prof_slow_paths.sh > gpurun_out/r4/slow_paths_kernels.txt 2>&1
  python bench.py > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err; tail -c 600 gpurun_out/r4/bench_final.json
  VPZ_BENCH_REHEARSAL=1 python bench.py --gpus 2 --single-process --steps 3 --warmup 1 > gpurun_out/r4/bench_single_process_2groups_one_gpu.json 2> gpurun_out/r4/bench_sp.err; tail -c 400 gpurun_out/r4/bench_single_process_2groups_one_gpu.json
  ;;
esac
ls gpurun_out/r4
