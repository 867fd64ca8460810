#!/usr/bin/env python3
"""debugging aid: where synth_big_kernel and the three-pass path (VPZ_NO_BIG=1) differ for one block-size pair"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import helpers
from vorbispizza_amd import Context, Decoder, capi, make_packets

size0, size1 = int(sys.argv[1]), int(sys.argv[2])
floored = len(sys.argv) < 4 or sys.argv[3] != "nofloor"
rng = np.random.default_rng(5)
channels, frames = 2, 12
bf = (rng.random(frames) < 0.6).astype(np.uint8) if size0 != size1 else np.zeros(frames, dtype=np.uint8)
prev = np.concatenate([[1], bf[:-1]]); nxt = np.concatenate([bf[1:], [1]])
flags = (bf * 1 | prev * 2 * bf | nxt * 4 * bf).astype(np.uint8)
h0, h1 = size0 // 2, size1 // 2
def xl(half, posts):
    return [0, half] + [int(v) for v in rng.choice(np.arange(1, half), size=posts - 2, replace=False)]
floors = [(xl(h0, 19), 2), (xl(h1, 29), 1)]
mappings = [{"coupling": [], "channel_floor": [0, 0]}, {"coupling": [], "channel_floor": [1, 1]}]
pk = make_packets(frames); parts = []; pp = []; cc = []; off = 0
for f in range(frames):
    half = h1 if bf[f] else h0
    res = (rng.standard_normal((channels, half)) * 3).round().astype(np.float32)
    posts, counts = helpers.random_posts(rng, floors[int(bf[f])][0], floors[int(bf[f])][1], channels, silent_prob=0.0)
    pk[f]["flags"] = int(flags[f]) | (0 if floored else capi.PKT_NO_FLOOR)
    if not floored: res = (res * 2.0 ** -6).astype(np.float32)
    pk[f]["mapping"], pk[f]["granule"], pk[f]["residue_offset"] = int(bf[f]), -1, off
    parts.append(res.reshape(-1)); pp.append(posts); cc.append(counts); off += res.size
res = np.concatenate(parts); posts = np.concatenate(pp).astype(np.int16); counts = np.concatenate(cc).astype(np.uint8)
ctx = Context(0)
def dec(no_big):
    if no_big: os.environ["VPZ_NO_BIG"] = "1"
    else: os.environ.pop("VPZ_NO_BIG", None)
    d = Decoder(ctx, channels, size0, size1, floors=floors, mappings=mappings)
    o = d.synth(pk, res, posts, counts)[0]; d.close(); return o
a, b = dec(True), dec(False)
print("block flags", bf.tolist(), "shape", a.shape, b.shape)
diff = np.abs(a - b)
print("max diff", diff.max(), "bit-equal", np.array_equal(a.view(np.uint32), b.view(np.uint32)))
# per frame output ranges
pos = 0
for f in range(1, frames):
    # samples of frame f: quarter sums
    prev_n, cur_n = (size1 if bf[f-1] else size0), (size1 if bf[f] else size0)
    cnt = prev_n // 4 + cur_n // 4
    seg = diff[:, pos:pos + cnt]
    bad = np.nonzero(seg.max(axis=0) > 0)[0]
    print("frame %2d (%s after %s) samples [%d,%d): max diff %.3g, differing samples %d%s" % (f, "L" if bf[f] else "s", "L" if bf[f-1] else "s", pos, pos + cnt,
          seg.max() if seg.size else 0, bad.size, (" first %d last %d" % (bad[0], bad[-1])) if bad.size else ""))
    pos += cnt
