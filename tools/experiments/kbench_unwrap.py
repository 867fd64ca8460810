#!/usr/bin/env python3
"""Tuning helper: configs[3]-shaped batches whose Floor1 has N posts, N in --posts; run under
`rocprofv3 --kernel-trace` and feed the trace to --parse to get the unwrap / synth kernel time per N."""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CALLS = 6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--posts", default="4,16,29")
    ap.add_argument("--parse", default=None, help="directory of a rocprofv3 --kernel-trace run of this script")
    args = ap.parse_args()
    ns = [int(v) for v in args.posts.split(",")]
    if args.parse:
        rows = []
        for path in glob.glob(args.parse + "/**/*kernel_trace.csv", recursive=True):
            with open(path, newline="") as f:
                for r in csv.DictReader(f):
                    if "vpz::" in r["Kernel_Name"]:
                        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                     r["Kernel_Name"].split("(")[0][-60:]))
        rows.sort()
        un = [d for _, d, n in rows if "unwrap" in n]
        sy = [d for _, d, n in rows if "synth_kernel" in n]
        for i, n in enumerate(ns):
            u = un[i * CALLS + 2:(i + 1) * CALLS]
            s = sy[i * CALLS + 2:(i + 1) * CALLS]
            print("posts %2d: unwrap %.1f us, synth %.1f us" % (n, sum(u) / len(u) / 1e3, sum(s) / len(s) / 1e3))
        return
    import numpy as np
    import torch
    import bench
    import helpers
    from vorbispizza_amd import Context, Decoder, capi
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    pk, res6, posts, counts, floors, mappings, samples6 = bench.build_floor6(torch, dev, 16384)
    out = torch.empty(6 * (samples6 + 1024), device=dev, dtype=torch.float32)
    cap = samples6 + 1024
    for n in ns:
        fl = [(helpers.LONG_XLIST[:n], 2)]
        cn = torch.full_like(counts, n)
        dec = Decoder(ctx, 6, 256, 2048, floors=fl, mappings=mappings)
        for _ in range(CALLS):
            dec.reset(-1)
            dec.synth_raw(pk, res6, posts, cn, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)
        ctx.synchronize()
        dec.close()
    ctx.close()


if __name__ == "__main__":
    main()
