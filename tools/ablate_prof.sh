#!/bin/bash
# Kernel durations (rocprofv3) with phases of the fused kernel switched off (VPZ_SYNTH_ABLATE: wrong results, right timing).
# usage: bash tools/ablate_prof.sh <which> "<list>"
set -e
W=${1:-real}
LIST=${2:-"0 16 8 2 1 3 27"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3/ablate_$W
mkdir -p $OUT
for ab in $LIST; do
  VPZ_SYNTH_ABLATE=$ab rocprofv3 --kernel-trace --stats -d "$OUT/ab$ab" -o k --output-format csv -- python tools/kbench_synth.py --steps 10 --which $W > "$OUT/ab$ab.log" 2>&1
  echo "ablate $ab: $(python tools/prof_summary.py $OUT/ab$ab/k_kernel_stats.csv | grep -v unwrap)"
done
