#!/usr/bin/env python3
"""configs[2], call by call: the GPU time of 40 consecutive vpz_decoder_synth calls on ONE fixed batch (HIP events around each
call) -- what the run cutting's determinism is judged by.  VPZ_HOST_PROFILE=1 adds the cut's parameters per call (stderr)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vorbispizza_amd import Context, Decoder, capi
which = sys.argv[1] if len(sys.argv) > 1 else "ola"
ctx = Context(0)
dev = torch.device("cuda", 0)
pk, residue, samples, res_floats = bench.build_synth_ola(torch, dev, 65536, all_long=(which == "olalong"))
dec = Decoder(ctx, 2, 256, 2048)
out = torch.empty(2 * (samples + 1024), device=dev, dtype=torch.float32)
cap = samples + 1024
us = []
for i in range(44):
    dec.reset(-1)
    ctx.timer_start()
    dec.synth_raw(pk, residue, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)
    us.append(ctx.timer_stop() * 1e3)
us = us[4:]
print("%s: 40 calls, GPU time per call: min %.1f mean %.1f max %.1f us (max/min %.3f)" % (which, min(us), sum(us) / len(us), max(us), max(us) / min(us)))
print(" ".join("%.0f" % u for u in us))
