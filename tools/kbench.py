#!/usr/bin/env python3
"""Kernel micro-benchmark used while tuning (not the contract bench): HIP-event time of
vpz_imdct_batch at BASELINE configs[1] size, repeated, median / min in microseconds."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=131072)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--mode", type=int, default=0)
    args = ap.parse_args()
    import torch
    from vorbispizza_amd import Context
    ctx = Context(0)
    x = torch.randn((args.count, args.n // 2), device="cuda") * 2.0 ** -8
    y = torch.empty((args.count, args.n), device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.imdct_batch(x, args.n, args.mode, out=y)
    ctx.synchronize()
    ts = []
    for _ in range(args.reps):
        ctx.timer_start()
        ctx.imdct_batch(x, args.n, args.mode, out=y)
        ts.append(ctx.timer_stop() * 1e3)
    ts = np.array(ts)
    byt = args.count * 6 * args.n
    print("imdct n=%d count=%d mode=%d: median %.1f us  min %.1f us  -> %.0f GB/s (median) %.0f GB/s (best)  env=%s"
          % (args.n, args.count, args.mode, np.median(ts), ts.min(), byt / np.median(ts) / 1e3, byt / ts.min() / 1e3,
             {k: v for k, v in os.environ.items() if k.startswith("VPZ_")}))


if __name__ == "__main__":
    main()
