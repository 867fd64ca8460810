#!/bin/bash
# Where does the fused kernel's time go?  Each phase switched off in turn (VPZ_SYNTH_ABLATE: wrong results, right timing).
# usage: bash tools/ablate.sh <which: real|ola|floor>
W=${1:-real}
for ab in 0 16 8 24 2 1 3 27; do
  echo "ablate $ab: $(VPZ_SYNTH_ABLATE=$ab python tools/kbench_synth.py --which $W --steps 40 2>&1 | grep configs)"
done
