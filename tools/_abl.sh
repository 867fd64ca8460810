cd /root/repo
timeout -k 10 600 python -m pytest tests/test_imdct_gpu.py tests/test_full_size_gpu.py tests/test_synth_gpu.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/kbench_imdct_sizes.py 2>&1 | grep "N ="
