#!/bin/bash
# SQ counters of the fused kernel for one decoder workload (run through gpurun; two rocprofv3 --pmc passes, eight
# counters each, no trace domains):  WHICH=floor|ola|real bash tools/pmc_sq_counters.sh   -> gpurun_out/pmc3.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc3a --output-format csv -- python3 $R/tools/kbench_synth.py --which ${WHICH:-floor} --steps 3 > $R/gpurun_out/pmc3a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc3b --output-format csv -- python3 $R/tools/kbench_synth.py --which ${WHICH:-floor} --steps 3 > $R/gpurun_out/pmc3b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3a synth_kernel > $R/gpurun_out/pmc3.txt; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3b synth_kernel >> $R/gpurun_out/pmc3.txt
