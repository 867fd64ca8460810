#!/bin/bash
# SQ counters of the fused kernel for one decoder workload (run through gpurun; rocprofv3 --pmc passes of eight
# counters each, no trace domains):  WHICH=floor|ola|real KERNEL=synth_dual_kernel bash tools/pmc_sq_counters.sh [out.txt]
# (BENCH=tools/kbench_slow_paths.py KERNEL=floor0_curve ... : another workload script, run without arguments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/pmc3.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $(dirname $OUT)
K=${KERNEL:-synth_}
if [ -n "$BENCH" ]; then ARGS="$R/$BENCH"; else ARGS="$R/tools/kbench_synth.py --which ${WHICH:-floor} --steps 3"; fi
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc3a --output-format csv -- python3 $ARGS > $R/gpurun_out/pmc3a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc3b --output-format csv -- python3 $ARGS > $R/gpurun_out/pmc3b.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INSTS_SMEM -d $R/gpurun_out/pmc3c --output-format csv -- python3 $ARGS > $R/gpurun_out/pmc3c.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3a $K > $OUT; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3b $K >> $OUT; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3c $K >> $OUT
cat $OUT
tail -3 $R/gpurun_out/pmc3a.log $R/gpurun_out/pmc3c.log | cut -c1-300
