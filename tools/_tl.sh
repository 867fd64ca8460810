cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl2
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl2 -o t --output-format csv -- python tools/kbench_synth.py --steps 6 --which ola > gpurun_out/tl2.log 2>&1
python tools/timeline.py gpurun_out/tl2 --last 14
