#!/bin/bash
# Kernel time alone (rocprofv3 --kernel-trace --stats) of the stereo fast path on north_star's workload with one phase switched
# off at a time (tuning build in vorbispizza_amd/lib_ab/tuning, VPZ_SYNTH_ABLATE: wrong results, right timing).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=${1:-olalong}
O=gpurun_out/r5_ablate_prof_$W
mkdir -p $O
export VPZ_LIB_DIR=$PWD/vorbispizza_amd/lib_ab/tuning
for ab in 0 64 1 2 4 32 1024 3 5 36 7 39; do
  export VPZ_SYNTH_ABLATE=$ab
  rocprofv3 --kernel-trace --stats -d $O/a$ab -o k --output-format csv -- python tools/kbench_synth.py --which $W --steps 10 > $O/a$ab.log 2>&1
  echo "ablate $ab: $(python tools/prof_summary.py $O/a$ab/k_kernel_stats.csv 2>&1 | grep synth_dual | head -1)" | tee -a $O/summary.txt
done
