cd /root/repo
for c in 128 256; do
echo "copies=$c merged"; timeout -k 10 200 python tools/kbench_synth.py --which real --copies $c --steps 10 2>&1 | tail -1
echo "copies=$c split"; VPZ_BENCH_SPLIT_SETUPS=1 timeout -k 10 200 python tools/kbench_synth.py --which real --copies $c --steps 10 2>&1 | tail -1
done
