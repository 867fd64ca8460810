#!/usr/bin/env python3
"""Round 5, review item 2 (the Floor1 curve out of the fused kernels' frame loop): what the curve would cost as a pass of its own --
floor1_render_kernel writing one byte per bin to memory for configs[3]'s 98 304 records and for the records of configs[4]'s
share -- run under rocprofv3 to read the kernel's time.  (The fused kernels' side of the balance is the tuning build's
VPZ_SYNTH_ABLATE=8: the same kernels without the curve, tools/experiments_ablate_prof.sh.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401  (the HIP runtime torch brings is the one the library binds to)
    import helpers
    from vorbispizza_amd import Context, Decoder
    which = sys.argv[1] if len(sys.argv) > 1 else "floor6"
    ctx = Context(0)
    if which == "floor6":
        frames, C6 = 16384, 6
        rng = np.random.default_rng(6)
        posts = np.zeros((frames * C6, 64), dtype=np.int16)
        posts[:, 0] = rng.integers(20, 60, size=frames * C6)
        posts[:, 1] = rng.integers(10, 40, size=frames * C6)
        v = rng.integers(0, 10, size=(frames * C6, 27))
        v[rng.random(v.shape) < 0.35] = 0
        posts[:, 2:29] = v
        counts = np.full(frames * C6, 29, dtype=np.uint8)
        dec = Decoder(ctx, C6, 256, 2048, floors=[(helpers.LONG_XLIST, 2)], mappings=[{"coupling": [], "channel_floor": [0] * C6}])
        rec_floor = np.zeros(frames * C6, dtype=np.uint8)
        rec_long = np.ones(frames * C6, dtype=np.uint8)
    else:
        from vorbispizza_amd.front import OggVorbisFile
        f = OggVorbisFile(os.path.join(ROOT, "tests", "golden", "3test.ogg"))
        pk, res, posts1, counts1 = f.decode_packets()
        copies = 96  # (about the records of configs[4]'s share: 128 streams of ~550 packets, two channels)
        posts = np.tile(posts1, (copies, 1))
        counts = np.tile(counts1, copies)
        dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
        long_pk = (pk["flags"] & 1).astype(np.uint8)
        mapping = pk["mapping"]
        rf = np.array([f.mappings[m]["channel_floor"][c] for m in mapping for c in range(f.channels)], dtype=np.uint8)
        rec_floor = np.tile(rf, copies)
        rec_long = np.tile(np.repeat(long_pk, f.channels), copies)
    for _ in range(3):
        curve, fy, fl, act = dec.debug_floor1_indices(posts, counts, rec_floor, rec_long)
    print("%s: %d records, curve bytes %d" % (which, len(counts), curve.size))
    dec.close()
    ctx.close()


if __name__ == "__main__":
    main()
