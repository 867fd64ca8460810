#!/usr/bin/env python3
"""Tuning helper: configs[4] end to end (host entropy decode on T threads + host-memory synth call) for
several thread counts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    from vorbispizza_amd import Context
    ctx = Context(0)
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for thr in (1, 2, 4, 8, 16):
        tot, (t_all, t_dec, t_syn) = bench.end_to_end_real_streams(ctx, torch, copies, thr)
        print("%2d threads: decode %.1f ms, synth(host mem) %.1f ms, end to end %.0f Msamples/s"
              % (thr, t_dec * 1e3, t_syn * 1e3, tot / t_all / 1e6), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
