#!/usr/bin/env python3
"""Tuning helper: vpz_imdct_batch (FAST) for every block size of the fused family, 1.6 GB of traffic per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from vorbispizza_amd import Context
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    for n in (256, 512, 1024, 2048, 4096, 8192):
        count = 131072 * 2048 // n
        x = torch.randn((count, n // 2), device=dev) * 2.0 ** -8
        y = torch.empty((count, n), device=dev)
        for _ in range(3):
            ctx.imdct_batch(x, n, out=y)
        ctx.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.imdct_batch(x, n, out=y)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / 20
            best = dt if best is None else min(best, dt)
        byt = 4 * (x.numel() + y.numel())
        print("N = %4d: %.3f ms/call  %.0f GB/s  (%.3f of 8 TB/s)" % (n, best * 1e3, byt / best / 1e9, byt / best / 8e12), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
