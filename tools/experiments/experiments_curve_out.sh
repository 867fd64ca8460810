#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_curve_out
mkdir -p $O
for w in floor6 real; do
  rocprofv3 --kernel-trace --stats -d $O/render_$w -o k --output-format csv -- python tools/experiments/experiments_curve_out.py $w > $O/render_$w.log 2>&1
  echo "== separate pass, $w: $(tail -1 $O/render_$w.log)" | tee -a $O/summary.txt
  python tools/prof_summary.py $O/render_$w/k_kernel_stats.csv 2>&1 | grep -E "floor1" | tee -a $O/summary.txt
done
export VPZ_LIB_DIR=$PWD/vorbispizza_amd/lib_ab/tuning
for w in floor real; do
  for ab in 0 8 256 264; do
    export VPZ_SYNTH_ABLATE=$ab
    rocprofv3 --kernel-trace --stats -d $O/${w}_a$ab -o k --output-format csv -- python tools/kbench_synth.py --which $w --steps 10 > $O/${w}_a$ab.log 2>&1
    echo "== fused kernel, $w, ablate $ab (8: no curve, 256: no floor multiply)" | tee -a $O/summary.txt
    python tools/prof_summary.py $O/${w}_a$ab/k_kernel_stats.csv 2>&1 | grep -E "synth|floor1" | tee -a $O/summary.txt
  done
done
