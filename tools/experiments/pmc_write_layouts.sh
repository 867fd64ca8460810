#!/bin/bash
# HBM bytes WRITTEN per dispatch of the stereo kernel on the configs[4] share, interleaved against planar PCM (rocprofv3 --pmc WRITE_SIZE,
# its own pass, no trace domains).  usage (through gpurun): bash tools/pmc_write_layouts.sh [out.txt]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/r4/pmc_write_layouts.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $(dirname $OUT); : > $OUT
for l in interleaved planar; do
  export VPZ_BENCH_REAL_LAYOUT=$l
  rm -rf $R/gpurun_out/pmcw_$l
  rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmcw_$l --output-format csv -- python3 $R/tools/kbench_synth.py --which real --steps 3 > $R/gpurun_out/pmcw_$l.log 2>&1
  echo "== PCM layout: $l" >> $OUT
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcw_$l synth_dual >> $OUT
done
cat $OUT
