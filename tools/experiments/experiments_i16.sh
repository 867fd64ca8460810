#!/bin/bash
# int16 residue read in place by the floored stereo kernel against the widen pass in front of it (VPZ_NO_DIRECT_I16=1):
# the 1 024-stream job through the dispatcher, three alternating rounds
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r5_i16_direct.txt}
: > $OUT
for round in 1 2 3; do
  for cfg in VPZ_NO_DIRECT_I16=1 VPZ_X=1; do
    echo "round $round $cfg: $(env $cfg python tools/dispatcher_probe.py 1,0,16,0,0 2>&1 | grep -v amdgpu | tail -2 | tr '\n' ' ' | cut -c1-300)" >> $OUT
  done
done
cat $OUT
