cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in merged split; do
  if [ $mode = split ]; then export VPZ_BENCH_SPLIT_SETUPS=1; unset VPZ_BENCH_MERGE_SETUPS; else export VPZ_BENCH_MERGE_SETUPS=1; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$mode --output-format csv -- python3 $R/tools/kbench_synth.py --which real --copies 512 --steps 4 > $R/gpurun_out/kt_$mode.log 2>&1
  echo $mode; tail -1 $R/gpurun_out/kt_$mode.log
  python3 - $R/gpurun_out/kt_$mode <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'vpz::' in r['Name']: print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
for f in glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'):
    seen=set()
    for r in csv.DictReader(open(f)):
        if 'synth_kernel' in r['Kernel_Name']:
            k=(r['Grid_Size_X'], r['Workgroup_Size_X'])
            if k not in seen: seen.add(k); print('   grid', k)
PY
done
