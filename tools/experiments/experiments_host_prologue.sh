#!/bin/bash
# Round 5: where a call's time goes at runs of 8 frames -- host phases (VPZ_HOST_PROFILE) and kernel time alone (rocprofv3),
# the whole kernel and its prologue only (tuning build, VPZ_SYNTH_ABLATE=64)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_host_prologue
mkdir -p $O
for cfg in "VPZ_DUAL_RUN=8" "VPZ_DUAL_RUN=32" "VPZ_NO_CHAIN=1"; do
  echo "== $cfg host phases" >> $O/summary.txt
  env $cfg VPZ_HOST_PROFILE=1 python tools/kbench_synth.py --which olalong --steps 6 2>&1 | grep "vpz host" | tail -4 >> $O/summary.txt
done
for cfg in "VPZ_DUAL_RUN=8" "VPZ_DUAL_RUN=32" "VPZ_NO_CHAIN=1"; do
  export $cfg
  rocprofv3 --kernel-trace --stats -d $O/k_$cfg -o k --output-format csv -- python tools/kbench_synth.py --which olalong --steps 10 > $O/k_$cfg.log 2>&1
  echo "== $cfg kernel stats" >> $O/summary.txt
  python tools/prof_summary.py $O/k_$cfg/k_kernel_stats.csv 2>&1 | head -5 >> $O/summary.txt
  unset ${cfg%%=*}
done
export VPZ_LIB_DIR=$PWD/vorbispizza_amd/lib_ab/tuning
for cfg in "VPZ_DUAL_RUN=8" "VPZ_DUAL_RUN=32"; do
  export $cfg VPZ_SYNTH_ABLATE=64
  rocprofv3 --kernel-trace --stats -d $O/p_$cfg -o k --output-format csv -- python tools/kbench_synth.py --which olalong --steps 10 > $O/p_$cfg.log 2>&1
  echo "== $cfg prologue only (ablate 64) kernel stats" >> $O/summary.txt
  python tools/prof_summary.py $O/p_$cfg/k_kernel_stats.csv 2>&1 | head -5 >> $O/summary.txt
  unset ${cfg%%=*}
done
cat $O/summary.txt
