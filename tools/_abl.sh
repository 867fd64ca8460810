cd /root/repo
VPZ_HOST_PROFILE=1 timeout -k 10 120 python tools/kbench_synth.py --which ola --steps 1 2>&1 | grep -v "^$" | tail -12
