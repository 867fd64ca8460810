#!/usr/bin/env python3
"""Tuning helper: what a SHORT block costs the fused kernel in group mode next to a long one -- stereo, Residue2-interleaved,
coupled, Floor1; one batch of long blocks only, one of short blocks only, same number of frames."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import helpers
    from vorbispizza_amd import Context, Decoder, capi, make_packets
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    streams, frames, C = 128, 480, 2
    rng = np.random.default_rng(1)
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}]
    for long_blocks in (True, False):
        half = 1024 if long_blocks else 128
        n = streams * frames
        pk = make_packets(n)
        fl = (capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG) if long_blocks else 0
        pk["flags"] = fl | capi.PKT_INTERLEAVED
        pk["mapping"] = 1 if long_blocks else 0
        pk["granule"] = -1
        pk["stream"] = np.repeat(np.arange(streams, dtype=np.int32), frames)
        pk["residue_offset"] = np.arange(n, dtype=np.int64) * (C * half)
        g = torch.Generator(device=dev).manual_seed(3)
        res = torch.round(torch.randn(n * C * half, generator=g, device=dev) * 4.0)
        xl = helpers.LONG_XLIST if long_blocks else helpers.SHORT_XLIST
        posts = np.zeros((n * C, 64), dtype=np.int16)
        posts[:, 0] = rng.integers(20, 60, size=n * C)
        posts[:, 1] = rng.integers(10, 40, size=n * C)
        posts[:, 2:len(xl)] = rng.integers(0, 8, size=(n * C, len(xl) - 2))
        counts = np.full(n * C, len(xl), dtype=np.uint8)
        d_posts, d_counts = torch.from_numpy(posts).to(dev), torch.from_numpy(counts).to(dev)
        dec = Decoder(ctx, C, 256, 2048, floors=floors, mappings=mappings, n_streams=streams)
        per = (frames - 1) * (half)
        cap = per + 2048
        out = torch.empty(streams * cap * C, device=dev, dtype=torch.float32)
        offs = np.arange(streams, dtype=np.int64) * cap * C

        def step():
            dec.reset(-1)
            dec.synth_raw(pk, res, d_posts, d_counts, out, offs, cap, capi.OUT_INTERLEAVED, 0, capi.MEM_DEVICE)

        for _ in range(3):
            step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%s blocks: %.3f ms per call of %d frames x %d ch  (%.2f ns per channel-frame)"
              % ("long " if long_blocks else "short", dt * 1e3, n, C, dt * 1e9 / (n * C)), flush=True)
        dec.close()
    ctx.close()


if __name__ == "__main__":
    main()
