#!/bin/bash
# the contract line of bench.py (fresh process, 5 warm-up + 20 timed steps, as the driver runs it) under round 4's cut and the chained
# short runs, alternating on one box; then the same with 200 timed steps
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r5_bench_ab.txt}
: > $OUT
for steps in 20 200; do
for round in 1 2 3; do
  for cfg in VPZ_NO_CHAIN=1 VPZ_X=1 VPZ_DUAL_RUN=16; do
    echo "steps $steps round $round $cfg: $(env $cfg python bench.py --steps $steps --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])')" >> $OUT
  done
done
done
cat $OUT
