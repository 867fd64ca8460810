#!/usr/bin/env python3
"""A/B of decoder settings INSIDE one process on one box (the process's "mode" -- DESIGN.md section 5 -- and the box are then the
same for every variant): one decoder per variant (a variant = environment variables the library reads at vpz_decoder_create),
the same device-resident batch, timed loops taken in turn, round after round; per variant the median and the best loop.

  python tools/ab_decoders.py --which olalong --variants "VPZ_NO_CHAIN=1" "VPZ_DUAL_RUN=8" "VPZ_DUAL_RUN=16"
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="olalong", choices=["olalong", "ola", "floor6", "real"])
    ap.add_argument("--variants", nargs="+", required=True, help='each "NAME=V[,NAME=V...]" or "default"')
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--layout", default="planar")
    ap.add_argument("--planar-in", action="store_true", help="floor6: the residue as planar [6][1024] packets instead of the Residue2 vector")
    ap.add_argument("--no-support", action="store_true", help="floor6: the mapping does not declare the residue's support")
    args = ap.parse_args()
    import torch
    import bench
    from vorbispizza_amd import Context, Decoder, capi
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    posts = counts = None
    floors, mappings, ch = (), (), 2
    offs = None
    expect = None
    if args.which == "real":  # configs[4]'s share of one GPU: 64 + 64 real stereo streams in one merged decoder, interleaved out
        from vorbispizza_amd import sharding
        parts = []
        for name, smp in bench.REAL_FIXTURES:
            f, pk1, res1, posts1, counts1, _ = bench.build_real_streams(torch, dev, name, 64)
            parts.append((f, pk1, res1, posts1, counts1, smp))
        floors, mappings, bases = sharding.merge_setups([(q[0].floors, q[0].mappings) for q in parts])
        pks, s0, r0 = [], 0, 0
        for (f, pk1, res1, posts1, counts1, smp), base in zip(parts, bases):
            pk1 = pk1.copy()
            pk1["stream"] += s0
            pk1["residue_offset"] += r0
            pk1["mapping"] += base
            pks.append(pk1)
            s0 += 64
            r0 += res1.numel()
        pk = np.concatenate(pks)
        residue = torch.cat([q[2] for q in parts])
        posts = torch.cat([q[3] for q in parts])
        counts = torch.cat([q[4] for q in parts])
        samples = max(q[5] for q in parts)
        expect = [q[5] for q in parts for _ in range(64)]
        byt = 8 * sum(expect) * 2
        args.layout = "interleaved"
    elif args.which == "floor6":
        pk, residue, posts, counts, floors, mappings, samples = bench.build_floor6(torch, dev, 16384, declare_support=not args.no_support)
        ch = 6
        if args.planar_in:
            residue = residue.reshape(16384, 1024, 6).transpose(1, 2).contiguous().reshape(-1)
            pk["flags"] &= np.uint8(~capi.PKT_INTERLEAVED & 0xFF)
        byt = 4 * residue.numel() // 2 + 4 * samples * 6 + posts.numel() * 2
    else:
        pk, residue, samples, res_floats = bench.build_synth_ola(torch, dev, args.frames, all_long=args.which == "olalong")
        byt = 4 * res_floats + 4 * samples * 2
    layout = capi.OUT_INTERLEAVED if args.layout == "interleaved" else capi.OUT_PLANAR
    decs = []
    for v in args.variants:
        keys = []
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("=")
                os.environ[k] = val
                keys.append(k)
        decs.append(Decoder(ctx, ch, 256, 2048, floors=floors, mappings=mappings, n_streams=128 if args.which == "real" else 1))
        for k in keys:
            del os.environ[k]
    cap = samples + 2048
    n_str = 128 if args.which == "real" else 1
    out = torch.empty(n_str * ch * cap, device=dev, dtype=torch.float32)
    if args.which == "real":
        offs = np.arange(n_str, dtype=np.int64) * cap * ch

    def loop(dec, n):
        for _ in range(n):
            dec.reset(-1)
            if args.which == "real":
                w = dec.synth_raw(pk, residue, posts, counts, out, offs, cap, layout, 0, capi.MEM_DEVICE, on_mismatch="ignore")
                assert [int(v) for v in w] == expect
            else:
                w = dec.synth_raw(pk, residue, posts, counts, out, None, cap, layout, cap, capi.MEM_DEVICE)
                assert int(w[0]) == samples

    for d in decs:
        loop(d, 3)
    ctx.synchronize()
    times = [[] for _ in decs]
    for r in range(args.rounds):
        for i, d in enumerate(decs):
            ctx.synchronize()
            ctx.timer_start()
            loop(d, args.steps)
            times[i].append(ctx.timer_stop() / args.steps)
    for v, t in zip(args.variants, times):
        t = np.array(t)
        print("%-40s median %.4f ms  best %.4f  worst %.4f  -> %.4f of 8 TB/s (median; %d bytes)"
              % (v, np.median(t), t.min(), t.max(), byt / (np.median(t) * 1e-3) / 8e12, byt))
    for d in decs:
        d.close()
    ctx.close()


if __name__ == "__main__":
    main()
