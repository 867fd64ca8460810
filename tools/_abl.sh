cd /root/repo
for rep in 1 2; do
VPZ_BENCH_SPLIT_SETUPS=1 timeout -k 10 200 python tools/kbench_synth.py --which real --steps 20 2>&1 | tail -1
timeout -k 10 200 python tools/kbench_synth.py --which real --steps 20 2>&1 | tail -1
done
