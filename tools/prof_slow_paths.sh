#!/bin/bash
# rocprofv3 kernel stats of tools/kbench_slow_paths.py (the workloads outside BASELINE's configs)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_slow_paths
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/k -o k --output-format csv -- python tools/kbench_slow_paths.py > $O/k.log 2>&1
grep "^(" $O/k.log
python tools/prof_summary.py $O/k/k_kernel_stats.csv 2>&1 | head -14
