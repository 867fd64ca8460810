cd /root/repo
for g in 1 2 3 4 5 6; do
export VPZ_IMDCT_GROUPS_PER_CU=$g
for i in 1 2; do
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('groups_per_cu $g', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
done; done
