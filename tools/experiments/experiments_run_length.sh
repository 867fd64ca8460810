#!/bin/bash
# Round 5, review item 3: what the run length (frames a wavefront walks before it jumps; VPZ_RUN_LENGTH overrides the
# fitted one) does to the fused stereo kernel.  Shorter runs = more, smaller regions streamed concurrently by round
# (tools/io_shapes.hip: 5.05 -> 5.5 -> 5.7 TB/s for runs of 32 / 16 / 8 frames with the arithmetic removed), paid for with one
# recomputed block per run.
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r5_run_length.txt}
: > $OUT
for rep in 1 2; do
for R in 0 32 24 16 12 8; do
  for w in olalong ola real; do
    if [ $R = 0 ]; then unset VPZ_RUN_LENGTH; else export VPZ_RUN_LENGTH=$R; fi
    echo "VPZ_RUN_LENGTH=${VPZ_RUN_LENGTH:-fitted} $w: $(python tools/kbench_synth.py --which $w --steps 40 2>&1 | tail -1)" >> $OUT
  done
done
done
cat $OUT
