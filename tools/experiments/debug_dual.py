"""Debug helper: where do the stereo fast path and the older routes differ?  Several calls per stream, device-resident
output poisoned before the first call; every route twice (is a route deterministic at all?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
import torch
from vorbispizza_amd import Context, Decoder, capi
from test_host_paths_gpu import env, stream_major_batch
ctx = Context(0)
dev = torch.device("cuda", 0)
n_streams, frames, channels = 18, 120, 2
cap = frames * 1024 + 64
splits = 3
for interleaved in (True, False):
    pk, res, _, _ = stream_major_batch(n_streams, frames, channels, seed=5200 + interleaved, floor=False, interleaved=interleaved, p_ls=0.2, p_sl=0.1)
    d_res = torch.from_numpy(res).to(dev)
    idx = np.arange(len(pk)).reshape(n_streams, frames)
    cuts = np.linspace(0, frames, splits + 1).astype(int)
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED):
      for host in (dict(VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3), dict(VPZ_PAR_MIN_PACKETS=1 << 40)):
        outs = {}
        for name, kv in (("dual", dict(VPZ_NO_DUAL=None)), ("dual2", dict(VPZ_NO_DUAL=None)), ("group", dict(VPZ_NO_DUAL=1)), ("group2", dict(VPZ_NO_DUAL=1))):
            with env(**dict(kv, **host)):
                junk = torch.full((n_streams * channels * cap,), float(len(outs) + 3), device=dev)  # stir the allocator
                dec = Decoder(ctx, channels, 256, 2048, n_streams=n_streams)
                out = torch.full((n_streams * channels * cap,), 7.0, device=dev)
                offs = np.arange(n_streams, dtype=np.int64) * channels * cap
                total = np.zeros(n_streams, dtype=np.int64)
                for a_, b_ in zip(cuts[:-1], cuts[1:]):
                    sub = pk[idx[:, a_:b_].reshape(-1)].copy()
                    w = dec.synth_raw(sub, d_res, None, None, out, offs + total * (channels if layout == capi.OUT_INTERLEAVED else 1),
                                      cap - int(total.max()), layout, cap, capi.MEM_DEVICE)
                    total += w
                ctx.synchronize()
                outs[name] = (out.cpu().numpy(), total.copy())
                dec.close()
                del junk
        for x, y in (("dual", "dual2"), ("group", "group2"), ("dual", "group")):
            a, b = outs[x][0], outs[y][0]
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            print("layout", layout, "ilv", interleaved, "host", "par" if len(host) == 2 else "ser", x, "vs", y, "differing:", len(bad), flush=True)
            if len(bad):
                i = int(bad[0])
                s, rem = divmod(i, channels * cap)
                ch, smp = divmod(rem, cap) if layout == capi.OUT_PLANAR else (rem % channels, rem // channels)
                print("  first diff: stream", s, "ch", ch, "sample", smp, "of", int(outs[x][1][s]), x, a[i], y, b[i])
                ss = bad[(bad >= s * channels * cap) & (bad < (s + 1) * channels * cap)] - s * channels * cap
                print("  bad in this stream:", len(ss), "first..last", int(ss[0]), int(ss[-1]), "streams touched:", sorted(set(int(v) // (channels * cap) for v in bad))[:20])
ctx.close()
