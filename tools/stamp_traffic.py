#!/usr/bin/env python3
"""Measures the HBM traffic of the headline kernel (imdct2048_kernel) with the PMC counters and stamps it.

Runs on the GPU box:  python tools/stamp_traffic.py
Two SEPARATE rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, HBM section) over
`bench.py --steps 5 --no-extras --no-cpu-baseline`, raw values in KiB, the gfx950 correction applied (FETCH_SIZE reads
exactly 1/2 of a wide coalesced streaming read; WRITE_SIZE is exact for 16-byte-per-lane streaming stores).  Writes
profiles/traffic_stamp.json with the date, the kernel name and the SHA-256 of the kernel's sources; bench.py puts the
stamped figure into `roofline.traffic` only while those sources are unchanged and prints null otherwise, so a kernel
edit cannot ride on an old measurement.
"""
import csv
import datetime
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "imdct2048_kernel"
KERNEL_SOURCES = ["vorbispizza_amd/csrc/imdct_fast.hip", "vorbispizza_amd/csrc/imdct_core.hpp"]
STAMP = os.path.join(ROOT, "profiles", "traffic_stamp.json")


def sources_digest():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def counter_pass(counter, outdir):
    shutil.rmtree(outdir, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "-d", outdir, "-o", "p", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--no-extras", "--no-cpu-baseline"]
    subprocess.run(cmd, check=True, cwd="/tmp", env=env, capture_output=True)
    vals = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise RuntimeError("no %s samples for %s" % (counter, KERNEL))
    return vals


def main():
    out = os.path.join(ROOT, "gpurun_out", "traffic_pmc")
    fetch = counter_pass("FETCH_SIZE", os.path.join(out, "fetch"))
    write = counter_pass("WRITE_SIZE", os.path.join(out, "write"))
    fetch_kib, write_kib = sum(fetch) / len(fetch), sum(write) / len(write)
    read_bytes = fetch_kib * 1024 * 2      # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes
    write_bytes = write_kib * 1024
    blocks = 65536 * 2
    algorithmic = blocks * (4 * 1024 + 4 * 2048)
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    stamp = {
        "kernel": KERNEL, "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "commit": commit or None,
        "kernel_sources": KERNEL_SOURCES, "kernel_sources_sha256": sources_digest(),
        "dispatches_sampled": [len(fetch), len(write)],
        "FETCH_SIZE_KiB_per_dispatch": round(fetch_kib, 1), "WRITE_SIZE_KiB_per_dispatch": round(write_kib, 1),
        "read_bytes_per_launch": int(read_bytes), "write_bytes_per_launch": int(write_bytes),
        "traffic_bytes_per_launch": int(read_bytes + write_bytes), "algorithmic_bytes_per_launch": algorithmic,
        "ratio_to_algorithmic": round((read_bytes + write_bytes) / algorithmic, 4),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 5 --no-extras "
                  "--no-cpu-baseline; KiB units; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
    }
    os.makedirs(os.path.dirname(STAMP), exist_ok=True)
    json.dump(stamp, open(STAMP, "w"), indent=1)
    print(json.dumps(stamp, indent=1))


if __name__ == "__main__":
    main()
