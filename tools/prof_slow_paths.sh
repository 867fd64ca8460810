#!/bin/bash
# rocprofv3 kernel stats of tools/kbench_slow_paths.py (the workloads outside BASELINE's configs)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r4/slowprof; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o k --output-format csv -- python3 tools/kbench_slow_paths.py > $D/log.txt 2>&1
python3 tools/prof_summary.py $D/k_kernel_stats.csv
