#!/bin/bash
# HBM traffic per dispatch of every library kernel of one tools/kbench_synth.py workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
# separate passes; KiB units, FETCH_SIZE x2 on gfx950 -- MI355X_MICROARCH.md).   usage: WHICH=floor bash tools/pmc_traffic.sh out.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
WHICH=${WHICH:-floor}
OUTF=${1:-gpurun_out/pmc_traffic_$WHICH.txt}
D=gpurun_out/pmc_traffic_tmp_$WHICH
rm -rf "$D" && mkdir -p "$D"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$D/$c" -o p --output-format csv -- python tools/kbench_synth.py --steps 3 --which $WHICH > "$D/$c.log" 2>&1
done
python - "$D" "$WHICH" > "$OUTF" <<'PY'
import csv, glob, sys, collections
d, which = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(d + "/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "vpz::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0][-70:]][c].append(float(r["Counter_Value"]))
print("# tools/kbench_synth.py --which %s: HBM traffic per dispatch (FETCH_SIZE x 2 KiB, WRITE_SIZE KiB; separate passes)" % which)
for k, cs in acc.items():
    rd = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])) * 1024 * 2
    wr = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])) * 1024
    print("%-70s read %8.1f MB  write %8.1f MB  (%d dispatches)" % (k, rd / 1e6, wr / 1e6, len(cs["FETCH_SIZE"])))
PY
cat "$OUTF"
