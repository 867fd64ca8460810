#!/usr/bin/env python3
"""Tuning helper (not the contract bench): runs the configs[2] / configs[3] decoder workloads a few
times; run it under `rocprofv3 --kernel-trace --stats` to read per-kernel times."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--frames6", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--which", default="both")
    ap.add_argument("--copies", type=int, default=64, help="streams of each fixture for --which real")
    ap.add_argument("--layout", default="planar", help="PCM layout of the ola / olalong workloads: planar (the bench's) or interleaved")
    args = ap.parse_args()
    import torch
    import bench
    from vorbispizza_amd import Context, Decoder, capi
    layout = capi.OUT_INTERLEAVED if args.layout == "interleaved" else None
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    if args.which in ("both", "ola"):
        pk, residue, samples, res_floats = bench.build_synth_ola(torch, dev, args.frames)
        dec = Decoder(ctx, 2, 256, 2048)
        dt, _ = bench.time_decoder(ctx, dec, torch, pk, residue, None, None, samples, 2, args.steps, 2, layout=layout)
        byt = 4 * res_floats + 4 * samples * 2
        print("configs[2]: %.3f ms/call  %.1f Msamples/s  %.0f GB/s algorithmic (whole call); algorithmic bytes %d"
              % (dt * 1e3, samples * 2 / dt / 1e6, byt / dt / 1e9, byt))
        dec.close()
        del residue
    if args.which == "olalong":  # north_star's literal line: all-long stereo IMDCT + window + OLA through the fused kernel
        pk, residue, samples, res_floats = bench.build_synth_ola(torch, dev, args.frames, all_long=True)
        dec = Decoder(ctx, 2, 256, 2048)
        dt, _ = bench.time_decoder(ctx, dec, torch, pk, residue, None, None, samples, 2, args.steps, 2, layout=layout)
        byt = 4 * res_floats + 4 * samples * 2
        print("north_star line: %.3f ms/call  %.1f Msamples/s  %.0f GB/s algorithmic (whole call); algorithmic bytes %d"
              % (dt * 1e3, samples * 2 / dt / 1e6, byt / dt / 1e9, byt))
        dec.close()
        del residue
    if args.which in ("both", "floor"):
        pk, res6, posts, counts, floors, mappings, samples6 = bench.build_floor6(torch, dev, args.frames6)
        dec = Decoder(ctx, 6, 256, 2048, floors=floors, mappings=mappings)
        dt, _ = bench.time_decoder(ctx, dec, torch, pk, res6, posts, counts, samples6, 6, args.steps, 2)
        byt = 4 * res6.numel() + 4 * samples6 * 6
        print("configs[3]: %.3f ms/call  %.1f Msamples/s  %.0f GB/s algorithmic (whole call)"
              % (dt * 1e3, samples6 * 6 / dt / 1e6, byt / dt / 1e9))
        dec.close()
    if args.which in ("both", "real"):
        dt, tot, _, _ = bench.time_real_streams(ctx, torch, dev, args.copies, steps=args.steps)
        print("configs[4] share (%d real stereo streams, interleaved out): %.3f ms/step  %.1f Msamples/s; algorithmic bytes %d"
              % (2 * args.copies, dt * 1e3, tot / dt / 1e6, 8 * tot))
    ctx.close()


if __name__ == "__main__":
    main()
