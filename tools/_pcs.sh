cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 -d $R/gpurun_out/pcs --output-format csv -- python3 $R/tools/kbench_synth.py --which floor --steps 20 > $R/gpurun_out/pcs.log 2>&1
echo rc=$? >> $R/gpurun_out/pcs.log
ls -la $R/gpurun_out/pcs/* >> $R/gpurun_out/pcs.log 2>&1
