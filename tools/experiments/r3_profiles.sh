#!/bin/bash
# Everything profiles/r3_* is made from, in one call on the GPU box:  bash tools/r3_profiles.sh
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
python tools/stamp_traffic.py > gpurun_out/r3/stamp_traffic.log 2>&1; tail -4 gpurun_out/r3/stamp_traffic.log
bash tools/collect_profiles.sh r3 > gpurun_out/r3/collect.log 2>&1; tail -30 gpurun_out/r3/collect.log
KERNEL=synth_dual WHICH=olalong bash tools/pmc_sq_counters.sh gpurun_out/r3/pmc_SQ_north_star_line.txt > /dev/null 2>&1
KERNEL=synth_dual WHICH=ola bash tools/pmc_sq_counters.sh gpurun_out/r3/pmc_SQ_config2.txt > /dev/null 2>&1
KERNEL=synth_dual WHICH=real bash tools/pmc_sq_counters.sh gpurun_out/r3/pmc_SQ_config4.txt > /dev/null 2>&1
KERNEL=synth_kernel WHICH=floor bash tools/pmc_sq_counters.sh gpurun_out/r3/pmc_SQ_config3.txt > /dev/null 2>&1
cp profiles/traffic_stamp.json gpurun_out/r3/traffic_stamp.json
ls gpurun_out/r3
