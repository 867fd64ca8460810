#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the summaries of tools/r4_profiles.sh under their round-4 names
cd "$(dirname "$0")/.."
P=gpurun_out/profiles_r4; R=gpurun_out/r4
cp $P/bench/b_kernel_stats.csv profiles/r4_bench_kernel_stats.csv
cp $P/bench_kernel_stats.txt profiles/r4_bench_kernel_stats.txt
grep '^{' $P/bench.json | tail -1 > profiles/r4_bench_under_rocprof.json
for pair in olalong:north_star_line ola:config2 floor:config3 real:config4; do
  w=${pair%%:*}; n=${pair##*:}
  cp $P/$w/k_kernel_stats.csv profiles/r4_kernel_stats_$n.csv
  cp $P/${w}_timeline.txt profiles/r4_timeline_$n.txt
done
cp $P/synth_path_traffic.txt profiles/r4_pmc_synth_path_traffic_config3.txt
for n in config2 config3 config4 north_star_line; do cp $R/pmc_SQ_$n.txt profiles/r4_pmc_SQ_$n.txt; done
cp $R/slow_paths.txt profiles/r4_slow_paths.txt
grep -v amdgpu.ids $R/slow_paths_kernels.txt >> profiles/r4_slow_paths.txt
grep '^{' $R/bench_final.json | tail -1 > profiles/r4_bench_final.json
grep '^{' $R/bench_single_process_2groups_one_gpu.json | tail -1 > profiles/r4_bench_single_process_2groups_one_gpu.json
cp $R/traffic_stamp.json profiles/traffic_stamp.json
ls -la profiles | grep -c r4_
