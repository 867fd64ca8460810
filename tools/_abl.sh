cd /root/repo
timeout -k 10 900 python -m pytest tests/test_host_paths_gpu.py tests/test_real_files_gpu.py tests/test_synth_gpu.py -x -q 2>&1 | tail -3
for a in 0 32 0; do
  echo "ABLATE=$a"; VPZ_SYNTH_ABLATE=$a timeout -k 10 120 python tools/kbench_synth.py --which floor --steps 40 2>&1 | tail -1
done
timeout -k 10 120 python tools/kbench_synth.py --which real --steps 40 2>&1 | tail -1
