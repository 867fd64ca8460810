#!/bin/bash
# The N = 2 path of bench.py rehearsed on a ONE-GPU box: both ranks on device 0, gloo for the barrier and the reductions (RCCL
# refuses two ranks on one GPU).  Never what the driver runs; checks that the partitioned job gives the single-GPU checksum.
# usage (through gpurun): bash tools/rehearse_n2.sh [out.json]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/bench_n2_rehearsal.json}
mkdir -p $(dirname $OUT)
VPZ_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 > $OUT 2> ${OUT%.json}.err || { tail -20 ${OUT%.json}.err; exit 1; }
python - "$OUT" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
job = [v for k, v in d["extra_workloads"].items() if "whole job" in k][0]
print("n_gpus", d["n_gpus"], "value", d["value"], "checksum", job["pcm_checksum"], job["checksum_equals_single_stream_decode"],
      "streams per rank", job["streams_per_rank"], "e2e", job["end_to_end_Msamples_per_s"], "s16", job["end_to_end_s16_Msamples_per_s"])
assert job["pcm_checksum"] == "b39d419c70046c00" and job["checksum_equals_single_stream_decode"]
PY
