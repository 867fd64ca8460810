#!/bin/bash
# The 8-way job rehearsed on a ONE-GPU box, both launchers (never what the driver runs):
#  (a) one process per rank under torchrun, gloo for the barrier / reductions, every rank on device 0.  The GPU boxes of this pool
#      allow at most SIX processes on the card at once (the process guard kills the run at a seventh, and torchrun's own agent
#      counts as one), so the per-process launcher is rehearsed with 5 ranks and --host-threads 2 -- the two host threads a
#      rank gets when 8 ranks share 16 CPUs --; the partition logic for 8 ranks is covered on the CPU
#      (tests/test_sharding_cpu.py) and (b) runs the full 8;
#  (b) ONE process, 8 context groups on device 0 through the in-process dispatcher (bench.py --gpus 8 --single-process).
# usage (through gpurun): bash tools/rehearse_n8.sh [outdir]
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r5_rehearsal}
mkdir -p $O
VPZ_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 5 --steps 5 --warmup 2 --host-threads 2 > $O/bench_n5_rehearsal_one_gpu.json 2> $O/bench_n5_rehearsal_one_gpu.err || { tail -20 $O/bench_n5_rehearsal_one_gpu.err; exit 1; }
VPZ_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 8 --single-process --steps 5 --warmup 2 > $O/bench_single_process_8groups_one_gpu.json 2> $O/bench_single_process_8groups_one_gpu.err || { tail -20 $O/bench_single_process_8groups_one_gpu.err; exit 1; }
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads([l for l in open(o + "/bench_n5_rehearsal_one_gpu.json") if l.startswith("{")][-1])
job = [v for k, v in d["extra_workloads"].items() if "whole job" in k][0]
print("torchrun: n_gpus", d["n_gpus"], "value", d["value"], "checksum", job["pcm_checksum"], job["checksum_equals_single_stream_decode"],
      "streams per rank", job["streams_per_rank"], "host threads per rank", job["host_threads_per_rank"], "e2e", job["end_to_end_Msamples_per_s"],
      "s16", job["end_to_end_s16_Msamples_per_s"])
assert job["pcm_checksum"] == "b39d419c70046c00" and job["checksum_equals_single_stream_decode"]
one = [v for k, v in d["extra_workloads"].items() if "ONE process" in k and "f32" in k][0]
print("  ... its one-process leg:", one["devices"], one["Msamples_per_s"], one["pcm_checksum"])
s = json.loads([l for l in open(o + "/bench_single_process_8groups_one_gpu.json") if l.startswith("{")][-1])
j8 = [v for k, v in s["extra_workloads"].items() if "f32" in k][0]
print("single process: n_gpus", s["n_gpus"], "value", s["value"], "devices", j8["devices"], "streams per device", j8["streams_per_device"],
      "checksum", j8["pcm_checksum"], "Msamples/s", j8["Msamples_per_s"], "threads per device", j8["host"].get("decode_threads_per_device") if "host" in j8 else None)
assert j8["pcm_checksum"] == "b39d419c70046c00"
PY
