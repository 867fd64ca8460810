#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the summaries of tools/r5_profiles.sh under their round-5 names
cd "$(dirname "$0")/.."
P=gpurun_out/profiles_r5; R=gpurun_out/r5
cp $R/contract_line/b_kernel_stats.csv profiles/r5_kernel_stats_north_star_line.csv
cp $R/contract_line_kernel_stats.txt profiles/r5_kernel_stats_north_star_line.txt
grep '^{' $R/contract_line.json | tail -1 > profiles/r5_bench_contract_line_under_rocprof.json
cp $P/bench/b_kernel_stats.csv profiles/r5_bench_kernel_stats.csv
cp $P/bench_kernel_stats.txt profiles/r5_bench_kernel_stats.txt
grep '^{' $P/bench.json | tail -1 > profiles/r5_bench_under_rocprof.json
for pair in ola:config2 floor:config3 real:config4; do
  w=${pair%%:*}; n=${pair##*:}
  cp $P/$w/k_kernel_stats.csv profiles/r5_kernel_stats_$n.csv
  cp $P/${w}_timeline.txt profiles/r5_timeline_$n.txt
done
cp $P/olalong_timeline.txt profiles/r5_timeline_north_star_line.txt
cp $P/synth_path_traffic.txt profiles/r5_pmc_synth_path_traffic_config3.txt
for n in config3 config4 north_star_line; do cp $R/pmc_SQ_$n.txt profiles/r5_pmc_SQ_$n.txt; done
grep "^(" $R/slow_paths.txt > profiles/r5_slow_paths.txt
grep -v amdgpu.ids $R/slow_paths_kernels.txt >> profiles/r5_slow_paths.txt
grep '^{' $R/bench_final.json | tail -1 > profiles/r5_bench_final.json
cp $R/traffic_stamp.json profiles/traffic_stamp.json
ls profiles | grep -c r5_
