#!/bin/bash
# The last two records of the round, made AFTER profiles/traffic_stamp.json is in place (the bench prints `traffic` only when the stamp's
# source hash matches): the contract line alone under rocprofv3, and the default bench.py run.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" &&
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/contract_line -o b --output-format csv -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r5/contract_line.json 2> gpurun_out/r5/contract_line.err
  python tools/prof_summary.py gpurun_out/r5/contract_line/b_kernel_stats.csv > gpurun_out/r5/contract_line_kernel_stats.txt; cat gpurun_out/r5/contract_line_kernel_stats.txt; tail -c 700 gpurun_out/r5/contract_line.json )
python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; tail -c 300 gpurun_out/r5/bench_final.json; wc -c gpurun_out/r5/bench_final.json
