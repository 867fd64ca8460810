cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in floor real; do
rm -rf gpurun_out/tl_$w
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl_$w -o t --output-format csv -- python tools/kbench_synth.py --steps 6 --which $w > gpurun_out/tl_$w.log 2>&1
echo "== $w"; python tools/timeline.py gpurun_out/tl_$w --last 10
done
