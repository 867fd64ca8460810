#!/usr/bin/env python3
"""Tuning helper: the configs[2] workload (one stereo stream, Markov window switching, pre-floored spectra) for
other block-size pairs -- the general variant of the fused kernel, or the three-pass path for sizes outside
{256, 512, 1024, 2048}."""
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
    pairs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(256, 2048), (512, 1024), (1024, 1024), (512, 4096)]
    for size0, size1 in pairs:
        frames = 65536 * 2048 // size1
        flags = helpers.markov_block_flags(frames, seed=3)
        if size0 == size1:
            flags &= ~np.uint8(7)
        halves = np.where(flags & 1, size1 // 2, size0 // 2).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(halves * 2)])
        pk = make_packets(frames)
        pk["flags"] = flags | capi.PKT_NO_FLOOR
        pk["granule"] = -1
        pk["residue_offset"] = offs[:-1]
        res = torch.randn(int(offs[-1]), device=dev) * 2.0 ** -8
        dec = Decoder(ctx, 2, size0, size1)
        cap = int(halves.sum()) + 4096
        out = torch.empty(2 * cap, device=dev)

        def step():
            dec.reset(-1)
            return dec.synth_raw(pk, res, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)

        for _ in range(2):
            w = step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5
        samples = int(w[0]) * 2
        print("%d/%d: %d frames, %.3f ms/call, %.1f Msamples/s, %.0f GB/s algorithmic"
              % (size0, size1, frames, dt * 1e3, samples / dt / 1e6, (4 * int(offs[-1]) + 4 * samples) / dt / 1e9), flush=True)
        dec.close()
        del res, out
    ctx.close()


if __name__ == "__main__":
    main()
