#!/bin/bash
# Instruction-cache counters of the fused kernels (is the 74 KB floor kernel fetch-bound?):
#   WHICH=real|ola|olalong|floor [VPZ_NO_DUAL=1] bash tools/pmc_icache.sh out.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/pmc_icache.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $(dirname $OUT)
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_ic --output-format csv -- python3 $R/tools/kbench_synth.py --which ${WHICH:-real} --steps 3 > $R/gpurun_out/pmc_ic.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_ic synth_ > $OUT
cat $OUT
