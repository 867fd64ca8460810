#!/bin/bash
# Round profiles on the GPU box (run through gpurun): kernel stats of the contract bench, kernel stats + device
# timelines + HBM traffic counters of the configs[2] / [3] / [4] decoder workloads.  Output under gpurun_out/profiles_rN/;
# the summaries worth keeping are copied to profiles/ by hand.   usage: bash tools/collect_profiles.sh r2
set -e
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/bench" -o b --output-format csv -- python bench.py --steps 20 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python tools/prof_summary.py "$OUT/bench/b_kernel_stats.csv" > "$OUT/bench_kernel_stats.txt"
for w in olalong ola floor real; do
  rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT/$w" -o k --output-format csv -- python tools/kbench_synth.py --steps 8 --which $w > "$OUT/$w.log" 2>&1
  python tools/prof_summary.py "$OUT/$w/k_kernel_stats.csv" > "$OUT/${w}_kernel_stats.txt"
  python tools/timeline.py "$OUT/$w" --last 12 --ours > "$OUT/${w}_timeline.txt"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o p --output-format csv -- python tools/kbench_synth.py --steps 3 --which floor > "$OUT/pmc_$c.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "vpz::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0][-60:]][c].append(float(r["Counter_Value"]))
with open(out + "/synth_path_traffic.txt", "w") as fh:
    fh.write("# configs[3] chain (6 ch, Residue2-interleaved, coupled, Floor1 on the GPU, 16384 frames): HBM traffic per dispatch\n")
    fh.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md)\n")
    tot = 0.0
    for k, cs in acc.items():
        rd = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])) * 1024 * 2
        wr = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])) * 1024
        tot += rd + wr
        fh.write("%-60s read %8.1f MB  write %8.1f MB\n" % (k, rd / 1e6, wr / 1e6))
    alg = 16384 * 6 * 1024 * 4 + 16383 * 6 * 1024 * 4 + 16384 * 6 * 64 * 2
    fh.write("total %.1f MB   algorithmic %.1f MB   ratio %.3f\n" % (tot / 1e6, alg / 1e6, tot / alg))
print(open(out + "/synth_path_traffic.txt").read())
PY
cat "$OUT"/*_kernel_stats.txt
