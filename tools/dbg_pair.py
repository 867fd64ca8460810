import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vorbispizza_amd import Context, Decoder, capi
from vorbispizza_amd.front import OggVorbisFile
ctx = Context(0)
f = OggVorbisFile(os.path.join(ROOT, "tests", "golden", "3test.ogg"))
n_streams = 4
parts = [f.decode_packets(stream_id=s) for s in range(n_streams)]
res_off = np.cumsum([0] + [p[1].size for p in parts])
pk = np.concatenate([p[0] for p in parts])
for s in range(n_streams):
    pk["residue_offset"][pk["stream"] == s] += res_off[s]
res = np.concatenate([p[1] for p in parts]); posts = np.concatenate([p[2] for p in parts]); counts = np.concatenate([p[3] for p in parts])
dec = Decoder(ctx, 2, 256, 2048, floors=f.floors, mappings=f.mappings, n_streams=n_streams)
for trial in range(3):
    dec.reset(-1)
    outs = dec.synth(pk, res, posts, counts, out_layout=capi.OUT_INTERLEAVED)
    pl = Decoder(ctx, 2, 256, 2048, floors=f.floors, mappings=f.mappings, n_streams=n_streams)
    ref = pl.synth(pk, res, posts, counts, out_layout=capi.OUT_PLANAR)
    pl.close()
    for s in range(n_streams):
        d = np.abs(outs[s] - ref[s].T)
        bad = np.argwhere(d > 0)
        print("trial", trial, "stream", s, "mismatches", len(bad), "max", d.max(), "first", bad[:3].tolist(), "last", bad[-3:].tolist())
# per-packet sample offsets for locating
ps = dec.last_packet_samples(len(pk))[: len(parts[0][0])]
cum = np.cumsum(ps)
print("packet boundaries near first mismatch:", [ (i,int(c)) for i,c in enumerate(cum[:12])])
ctx.close()
