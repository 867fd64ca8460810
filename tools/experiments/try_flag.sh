#!/bin/bash
# Tuning: the contract line (bench.py --no-extras) with an extra compile flag against the product build, alternating on one box.
# usage: bash tools/try_flag.sh -DVPZ_SOMETHING
F=$1
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
echo "product : $(run)"
VPZ_EXTRA_HIPCC_FLAGS=$F python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
export VPZ_EXTRA_HIPCC_FLAGS=$F
echo "$F : $(run)"; echo "$F : $(run)"
unset VPZ_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
echo "product : $(run)"
