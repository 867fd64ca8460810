cd /root/repo
timeout -k 10 900 python -m pytest tests/test_floor1_integers_gpu.py tests/test_host_paths_gpu.py tests/test_synth_gpu.py tests/test_real_files_gpu.py -x -q 2>&1 | tail -3 &&
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/ktu --output-format csv -- python3 $R/tools/kbench_unwrap.py --posts 2,4,16,29 > $R/gpurun_out/ktu.log 2>&1
python3 $R/tools/kbench_unwrap.py --posts 2,4,16,29 --parse $R/gpurun_out/ktu
