#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel (name prefix) and counter: `pmc_summary.py DIR [substr]`."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "vpz::"
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if needle not in name:
                continue
            short = name.split("(")[0][-60:]
            a = acc[short][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print("   %-28s %16.0f per dispatch (%d dispatches)" % (c, v / n, n))
