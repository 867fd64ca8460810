#!/bin/bash
# Kernel-level A/B (rocprofv3 --kernel-trace --stats) of one kbench_synth workload with and without the stereo fast path.
# usage: bash tools/prof_ab.sh <which: real|ola> <outdir>
set -e
W=${1:-real}
OUT=${2:-gpurun_out/r3/prof_ab}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for nd in 0 1; do
  export VPZ_NO_DUAL=$nd
  rocprofv3 --kernel-trace --stats -d "$OUT/${W}_nodual$nd" -o k --output-format csv -- python tools/kbench_synth.py --steps 10 --which $W > "$OUT/${W}_nodual$nd.log" 2>&1
  echo "== $W VPZ_NO_DUAL=$nd"
  python tools/prof_summary.py "$OUT/${W}_nodual$nd/k_kernel_stats.csv"
done
