cd /root/repo
for rep in 1 2 3; do
for a in 0 512; do
  echo "ABLATE=$a"; VPZ_SYNTH_ABLATE=$a timeout -k 10 120 python tools/kbench_synth.py --which floor --steps 40 2>&1 | tail -1
done
done
