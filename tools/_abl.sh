cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 &&
timeout -k 10 300 python tools/kbench_layouts.py 2>&1 | grep "ms/call"
