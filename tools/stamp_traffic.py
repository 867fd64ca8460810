#!/usr/bin/env python3
"""Measures the HBM traffic of the headline kernel (imdct2048_kernel) and of the fused kernels with the PMC counters and
stamps it.

Runs on the GPU box:  python tools/stamp_traffic.py
Two SEPARATE rocprofv3 passes per workload (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, HBM
section), raw values in KiB, the gfx950 correction applied (FETCH_SIZE reads exactly 1/2 of a wide coalesced streaming
read; WRITE_SIZE is exact for 16-byte-per-lane streaming stores).  Writes profiles/traffic_stamp.json with the date, the
kernel names and the SHA-256 of each kernel's sources; bench.py puts a stamped figure into a `roofline.traffic` only while
those sources are unchanged and prints null otherwise, so a kernel edit cannot ride on an old measurement.
"""
import csv
import datetime
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.path.join(ROOT, "profiles", "traffic_stamp.json")
FUSED_SOURCES = ["vorbispizza_amd/csrc/synth_dual.hip", "vorbispizza_amd/csrc/synth_common.hpp",
                 "vorbispizza_amd/csrc/imdct_core.hpp", "vorbispizza_amd/csrc/synth_desc.hpp",
                 # (the run cutting decides how many blocks a launch recomputes: a change there changes the bytes moved)
                 "vorbispizza_amd/csrc/vpz_decoder.hip"]
WORKLOADS = {
    # key: (command after the interpreter, kernel-name substrings, sources, algorithmic bytes or None = parse the log)
    "headline": (["tools/kbench.py", "--reps", "5"], ["imdct2048_kernel"],
                 ["vorbispizza_amd/csrc/imdct_fast.hip", "vorbispizza_amd/csrc/imdct_core.hpp"], 65536 * 2 * (4 * 1024 + 4 * 2048)),
    "north_star_line": (["tools/kbench_synth.py", "--which", "olalong", "--steps", "3"], ["synth_dual_kernel<false, false, 0, false"],
                        FUSED_SOURCES, None),
    "configs2": (["tools/kbench_synth.py", "--which", "ola", "--steps", "3"], ["synth_dual_kernel<false, false, 0, false"],
                 FUSED_SOURCES, None),
    "configs4_share": (["tools/kbench_synth.py", "--which", "real", "--steps", "3"],
                       ["synth_dual_kernel<true, true, 1, false", "floor1_unwrap_kernel"], FUSED_SOURCES + ["vorbispizza_amd/csrc/synth_kernels.hip"], None),
}


def sources_digest(sources):
    h = hashlib.sha256()
    for rel in sources:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def counter_pass(counter, outdir, cmd, kernels):
    shutil.rmtree(outdir, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--pmc", counter, "-d", outdir, "-o", "p", "--output-format", "csv", "--", sys.executable,
            os.path.join(ROOT, cmd[0])] + cmd[1:]
    r = subprocess.run(full, check=True, cwd="/tmp", env=env, capture_output=True, text=True)
    per_kernel = {k: [] for k in kernels}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            for k in kernels:
                if k in row["Kernel_Name"]:
                    per_kernel[k].append(float(row["Counter_Value"]))
    if not all(per_kernel.values()):
        raise RuntimeError("no %s samples for %r" % (counter, [k for k, v in per_kernel.items() if not v]))
    return per_kernel, r.stdout


def main():
    out = os.path.join(ROOT, "gpurun_out", "traffic_pmc")
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    stamp = {"date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "commit": commit or None,
             "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes per workload; KiB units; FETCH_SIZE x2 on "
                       "gfx950 (MI355X_MICROARCH.md, HBM section); per launch = mean over the sampled dispatches, summed over the "
                       "workload's kernels", "workloads": {}}
    only = sys.argv[1:] or list(WORKLOADS)
    for key in only:
        cmd, kernels, sources, algorithmic = WORKLOADS[key]
        fetch, log = counter_pass("FETCH_SIZE", os.path.join(out, key + "_fetch"), cmd, kernels)
        write, _ = counter_pass("WRITE_SIZE", os.path.join(out, key + "_write"), cmd, kernels)
        if algorithmic is None:
            algorithmic = int(re.search(r"algorithmic bytes (\d+)", log).group(1))
        read_bytes = sum(sum(v) / len(v) for v in fetch.values()) * 1024 * 2  # gfx950: 128-byte requests tallied at 64 bytes
        write_bytes = sum(sum(v) / len(v) for v in write.values()) * 1024
        stamp["workloads"][key] = {
            "kernels": kernels, "kernel_sources": sources, "kernel_sources_sha256": sources_digest(sources),
            "dispatches_sampled": {k: [len(fetch[k]), len(write[k])] for k in kernels},
            "read_bytes_per_launch": int(read_bytes), "write_bytes_per_launch": int(write_bytes),
            "traffic_bytes_per_launch": int(read_bytes + write_bytes), "algorithmic_bytes_per_launch": int(algorithmic),
            "ratio_to_algorithmic": round((read_bytes + write_bytes) / algorithmic, 4)}
        print(key, json.dumps(stamp["workloads"][key]), flush=True)
    # the headline entry keeps its old top-level fields too (bench.py of earlier rounds read them)
    if "headline" in stamp["workloads"]:
        h = stamp["workloads"]["headline"]
        stamp.update({"kernel": "imdct2048_kernel", "kernel_sources": h["kernel_sources"],
                      "kernel_sources_sha256": h["kernel_sources_sha256"], "traffic_bytes_per_launch": h["traffic_bytes_per_launch"],
                      "algorithmic_bytes_per_launch": h["algorithmic_bytes_per_launch"], "ratio_to_algorithmic": h["ratio_to_algorithmic"]})
    os.makedirs(os.path.dirname(STAMP), exist_ok=True)
    json.dump(stamp, open(STAMP, "w"), indent=1)
    # (gpurun brings back only gpurun_out/: leave a copy there, to be copied over profiles/traffic_stamp.json)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(stamp, open(os.path.join(ROOT, "gpurun_out", "traffic_stamp.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
