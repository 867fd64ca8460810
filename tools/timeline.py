#!/usr/bin/env python3
"""Prints the device timeline (kernels + memory copies) of the LAST `--calls` synth calls found in a
rocprofv3 output directory (--kernel-trace --memory-copy-trace --output-format csv), with the idle gap in
front of every entry.  Tuning helper."""
import argparse
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last", type=int, default=24)
    ap.add_argument("--ours", action="store_true", help="only the library's kernels and the memory copies")
    args = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    for f in glob.glob(os.path.join(args.dir, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
    rows.sort()
    if args.ours:
        rows = [r for r in rows if "vpz::" in r[2] or r[2].startswith("COPY") or "fillBuffer" in r[2]]
        while rows and "vpz::" not in rows[-1][2]:  # (what follows the last call -- checksums, copies back -- is not ours)
            rows.pop()
    rows = rows[-args.last:]
    prev_end = rows[0][0]
    for s, e, name in rows:
        print("%9.1f us gap  %9.1f us  %s" % ((s - prev_end) / 1e3, (e - s) / 1e3, name))
        prev_end = max(prev_end, e)
    print("span %.1f us" % ((rows[-1][1] - rows[0][0]) / 1e3))


if __name__ == "__main__":
    main()
