cd /root/repo
export VPZ_BENCH_REHEARSAL=1
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/reh.err | tail -1 > gpurun_out/reh.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/reh.json').read())
print(d['n_gpus'], d['value'], d['ms_per_step'])
for k,v in d['extra_workloads'].items():
    if 'whole job' in k: print(k[:60], json.dumps(v)[:900])
PY
