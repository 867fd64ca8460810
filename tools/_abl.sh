cd /root/repo
timeout -k 10 900 python -m pytest tests/test_floor1_integers_gpu.py tests/test_host_paths_gpu.py tests/test_synth_gpu.py tests/test_real_files_gpu.py tests/test_golden_vectors_gpu.py -x -q 2>&1 | tail -15 &&
timeout -k 10 200 python tools/kbench_synth.py --steps 5 2>&1 | grep -v "^$" | tail -3
