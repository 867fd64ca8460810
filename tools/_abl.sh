cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 &&
timeout -k 10 200 python tools/kbench_synth.py --steps 40 2>&1 | grep -v "^$" | tail -3
