#!/usr/bin/env python3
"""configs[2]: does the time of a FIXED batch depend on where the caller's buffers lie?  The same batch, the same decoder, 40
back-to-back calls per trial (wall time, one synchronisation at the end); between trials only the addresses of the residue and PCM
buffers change (a pad in front of each).  And: the same trial repeated without changing anything."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vorbispizza_amd import Context, Decoder, capi
which = sys.argv[1] if len(sys.argv) > 1 else "ola"
ctx = Context(0)
dev = torch.device("cuda", 0)
pk, residue0, samples, res_floats = bench.build_synth_ola(torch, dev, 65536, all_long=(which == "olalong"))
cap = samples + 1024
dec = Decoder(ctx, 2, 256, 2048)

def trial(residue, out):
    for _ in range(3):
        dec.reset(-1); dec.synth_raw(pk, residue, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            dec.reset(-1); dec.synth_raw(pk, residue, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    return best * 1e3

out0 = torch.empty(2 * cap, device=dev, dtype=torch.float32)
print("same buffers, five times:", " ".join("%.4f" % trial(residue0, out0) for _ in range(5)))
for pad_r, pad_o in ((0, 0), (4, 0), (0, 4), (1024, 0), (0, 1024), (65536, 0), (0, 65536), (1 << 20, 1 << 19), (12345 * 4, 777 * 4)):
    rbuf = torch.empty(res_floats + pad_r, device=dev, dtype=torch.float32)
    r = rbuf[pad_r:]
    r.copy_(residue0)
    obuf = torch.empty(2 * cap + pad_o, device=dev, dtype=torch.float32)
    o = obuf[pad_o:]
    print("residue +%8d floats (%x), pcm +%8d floats (%x): %.4f ms" % (pad_r, r.data_ptr() & 0xFFFFFF, pad_o, o.data_ptr() & 0xFFFFFF, trial(r, o)))
    del rbuf, obuf, r, o
