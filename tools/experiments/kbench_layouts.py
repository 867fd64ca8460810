#!/usr/bin/env python3
"""Tuning helper: the instantiations of the fused kernel that the contract bench does not time -- interleaved output
for channel counts other than 2 (`ReadSamples(Span<float>)` of a mono or 5.1 stream), with and without the floor /
group mode.  Whole vpz_decoder_synth calls, device-resident inputs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    from vorbispizza_amd import Context, Decoder, capi
    ctx = Context(0)
    dev = torch.device("cuda", 0)

    def run(label, dec, pk, res, posts, counts, samples, channels, layout):
        cap = samples + 1024
        out = torch.empty(channels * cap, device=dev, dtype=torch.float32)

        def step():
            dec.reset(-1)
            return dec.synth_raw(pk, res, posts, counts, out, None, cap, layout, cap, capi.MEM_DEVICE)

        for _ in range(2):
            step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5
        byt = 4 * res.numel() + 4 * samples * channels
        print("%-62s %.3f ms/call  %8.1f Msamples/s  %5.0f GB/s algorithmic" % (label, dt * 1e3, samples * channels / dt / 1e6,
                                                                             byt / dt / 1e9), flush=True)
        dec.close()

    # mono, mixed 256/2048, no floor: planar vs interleaved (synth_kernel<false, 0 / 1, false, false>)
    pk, res, samples, _ = bench.build_synth_ola(torch, dev, 65536)
    pk1 = pk.copy()
    pk1["residue_offset"] //= 2
    res1 = res[: res.numel() // 2].contiguous()
    for layout, name in ((capi.OUT_PLANAR, "planar"), (capi.OUT_INTERLEAVED, "interleaved")):
        run("mono 256/2048 window switching, %s" % name, Decoder(ctx, 1, 256, 2048), pk1, res1, None, None, samples, 1, layout)
    # 6 channels, Residue2-interleaved + coupling + Floor1 (group mode): planar vs interleaved output
    pk6, res6, posts, counts, floors, mappings, samples6 = bench.build_floor6(torch, dev, 16384)
    for layout, name in ((capi.OUT_PLANAR, "planar"), (capi.OUT_INTERLEAVED, "interleaved")):
        run("6 ch coupled + Floor1 (group mode), %s" % name, Decoder(ctx, 6, 256, 2048, floors=floors, mappings=mappings),
            pk6, res6, posts, counts, samples6, 6, layout)
    # ... and the same stream handed over PLANAR ([channel][bin] per packet): no de-interleave -- the channel-pair path of
    # synth_dual_kernel reads contiguous rows (VPZ_NO_DUAL=1: group mode)
    pk6p = pk6.copy()
    pk6p["flags"] &= ~np.uint8(capi.PKT_INTERLEAVED)
    res6p = res6.view(16384, 1024, 6).permute(0, 2, 1).contiguous().view(-1)
    run("6 ch coupled + Floor1, planar INPUT, planar out", Decoder(ctx, 6, 256, 2048, floors=floors, mappings=mappings),
        pk6p, res6p, posts, counts, samples6, 6, capi.OUT_PLANAR)
    ctx.close()


if __name__ == "__main__":
    main()
