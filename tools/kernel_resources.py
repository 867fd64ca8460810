#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS report of the HIP translation units, from the compiler's own
`-Rpass-analysis=kernel-resource-usage` remarks (no GPU needed).  tests/test_kernel_resources_cpu.py uses it to
fail the build check when a hot kernel instantiation spills to scratch.

  python tools/kernel_resources.py [file.hip ...]      (default: every translation unit of the library)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds", "SGPRs Spill": "sgpr_spill",
          "VGPRs Spill": "vgpr_spill"}


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), text=True,
                             capture_output=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def analyse(source, extra=()):
    """Returns [{name, demangled, vgprs, sgprs, scratch, occupancy, lds, ...}] for one .hip file."""
    from vorbispizza_amd import _build
    cmd = [_build._hipcc()] + _build.COMMON + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", source,
                                                             "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: (.*?)\s*\[-Rpass-analysis", line) or \
            re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            kernels.append(cur)
            continue
        if cur is None or ":" not in text:
            continue
        key, val = (t.strip() for t in text.rsplit(":", 1))
        if key in FIELDS:
            try:
                cur[FIELDS[key]] = int(val)
            except ValueError:
                pass
    names = demangle([k["name"] for k in kernels])
    for k in kernels:
        k["demangled"] = names.get(k["name"], k["name"])
    return kernels


def all_units():
    from vorbispizza_amd import _build
    return [(os.path.join(_build.CSRC, n), e) for n, e in _build.SOURCES.items()]


def main():
    units = [(a, ()) for a in sys.argv[1:]] or all_units()
    for src, extra in units:
        print("== %s" % os.path.relpath(src, ROOT))
        for k in analyse(src, extra):
            print("  %-92s vgpr %3d  scratch %4d  lds %6d  occ %d" % (
                re.sub(r"\(.*", "", k["demangled"])[:92], k.get("vgprs", -1), k.get("scratch", -1), k.get("lds", -1),
                k.get("occupancy", -1)))


if __name__ == "__main__":
    main()
