#!/bin/bash
# The whole GPU suite once per alternative route (the A/B switches force the routes the defaults do not take on a box like this).
# usage (through gpurun): bash tools/suite_under_switches.sh [out.txt]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/suite_under_switches.txt}
mkdir -p $(dirname $OUT)
for kv in ${SWITCHES:-VPZ_NO_DUAL=1 VPZ_NO_GROUP=1 VPZ_NO_COMPACT=1 VPZ_NO_EARLY_UPLOAD=1 VPZ_HOST_THREADS=3 VPZH_NO_SETUP_CACHE=1 VPZ_NO_SUPPORT=1 VPZ_NO_F0_FUSED=1}; do
  echo "== $kv"
  env $kv timeout -k 10 300 python -m pytest tests -q -m gpu -x 2>&1 | tail -1 || exit 1
done 2>&1 | tee $OUT
