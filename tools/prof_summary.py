#!/usr/bin/env python3
"""Prints the vpz:: rows of a rocprofv3 *_kernel_stats.csv."""
import csv
import sys

for row in csv.DictReader(open(sys.argv[1])):
    if "vpz::" in row["Name"]:
        print("%-60s calls %4s  avg %9.1f us  min %9.1f us" % (row["Name"].split("(")[0][-60:], row["Calls"],
              float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
