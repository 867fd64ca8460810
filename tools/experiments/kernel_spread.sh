#!/bin/bash
# Per-dispatch durations of the fused kernel over consecutive identical calls (rocprofv3 --kernel-trace), for configs[2] (ola) or
# another kbench_synth workload, under an environment switch: what the spread of a FIXED batch is made of.
#   usage: bash tools/kernel_spread.sh <out.txt> <which> [ENV=VALUE ...]
OUT=$1; W=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r4/kspread_$$
rm -rf $D; mkdir -p $D
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d $D -o k --output-format csv -- python3 tools/kbench_synth.py --which $W --steps 40 > $D/log.txt 2>&1
python3 - "$D" "$W" "$*" >> "$OUT" <<'PY'
import csv, glob, sys
d, w, env = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/k_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "synth_dual_kernel" in r["Kernel_Name"] or "synth_kernel" in r["Kernel_Name"]]
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
us = us[-120:]
import statistics
print("%s [%s]: %d dispatches, min %.1f mean %.1f max %.1f us, sigma %.1f, max/min %.3f" % (w, env, len(us), min(us), statistics.mean(us), max(us), statistics.pstdev(us), max(us) / min(us)))
print("  " + " ".join("%.0f" % u for u in us[-60:]))
PY
tail -2 "$OUT"
