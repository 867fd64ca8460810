#!/usr/bin/env python3
"""The paths no BASELINE config exercises, timed once so that BASELINE.md can say what they cost:
  (a) more than 8 channels (no group mode: separate coupling pass through a planar temp), 10 channels, Floor1 + coupling;
  (b) a type-0 floor stream (Floor0.Apply on the planar temp, then the fused kernel), stereo;
  (c) long blocks of 4096 / 8192 samples: 512/4096 and 1024/8192 (synth_big_kernel since round 5; VPZ_NO_BIG=1: the three-pass path);
whole vpz_decoder_synth calls, device-resident inputs, algorithmic bytes = residue in + PCM out."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(ctx, step, reps=20):
    # (20 calls after 3: with 5 after 2 -- rounds 2 and 3 -- a fifth of the figure was the first calls' launch gaps and the one
    # synchronisation, 0.296 ms where the device timeline shows a call every 0.260)
    for _ in range(3):
        w = step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps, w


def main():
    import torch
    import helpers
    from vorbispizza_amd import Context, Decoder, capi, make_packets
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)

    # (a) 10 channels, Residue2-interleaved, coupling (0,1),(2,3),(4,5), 29-post Floor1, all long blocks
    C_, frames = 10, 8192
    pk = make_packets(frames)
    pk["flags"] = capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG | capi.PKT_INTERLEAVED
    pk["granule"] = -1
    pk["residue_offset"] = np.arange(frames, dtype=np.int64) * (1024 * C_)
    res = torch.round(torch.randn((frames, 1024, C_), device=dev) * 4.0)
    res[:, 410:, :] = 0
    posts = np.zeros((frames * C_, 64), dtype=np.int16)
    posts[:, 0] = rng.integers(20, 60, size=frames * C_)
    posts[:, 1] = rng.integers(10, 40, size=frames * C_)
    posts[:, 2:29] = rng.integers(0, 8, size=(frames * C_, 27))
    counts = np.full(frames * C_, 29, dtype=np.uint8)
    d_posts, d_counts = torch.from_numpy(posts).to(dev), torch.from_numpy(counts).to(dev)
    planar_a = os.environ.get("SLOW_A_PLANAR", "0") not in ("", "0")  # (a) with planar [10][1024] packets (residue types 0 / 1)
    if planar_a:
        res = res.transpose(1, 2).contiguous()
        pk["flags"] &= np.uint8(~capi.PKT_INTERLEAVED & 0xFF)
    dec = Decoder(ctx, C_, 256, 2048, floors=[(helpers.LONG_XLIST, 2)],
                  mappings=[{"coupling": [(0, 1), (2, 3), (4, 5)], "channel_floor": [0] * C_}])
    samples = (frames - 1) * 1024
    cap = samples + 2048
    out = torch.empty(C_ * cap, device=dev)

    def step_a():
        dec.reset(-1)
        return dec.synth_raw(pk, res.view(-1), d_posts, d_counts, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)

    dt, w = timed(ctx, step_a)
    byt = 4 * res.numel() + 4 * samples * C_
    print("(a) 10 channels, %s, coupled, Floor1, %d frames (Residue2 vector: separate coupling pass + one wave per channel; planar packets: "
          "the pair route, VPZ_NO_PAIRS=1 the separate pass): %.3f ms/call  %.1f Msamples/s  "
          "%.0f GB/s algorithmic = %.3f of 8 TB/s" % ("planar packets" if planar_a else "Residue2 vector", frames, dt * 1e3, samples * C_ / dt / 1e6, byt / dt / 1e9, byt / dt / 8e12), flush=True)
    dec.close()
    del res, out

    # (b) stereo, type-0 floor (order 16), all long blocks, no coupling
    C_, frames = 2, 32768
    f0 = {"order": 16, "rate": 44100, "bark_map_size": 256, "amp_bits": 6, "amp_ofs": 40}
    pk = make_packets(frames)
    pk["flags"] = capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG
    pk["granule"] = -1
    pk["residue_offset"] = np.arange(frames, dtype=np.int64) * (1024 * C_)
    res = torch.round(torch.randn(frames * 1024 * C_, device=dev) * 4.0)
    amp = rng.uniform(1.0, 50.0, size=frames * C_).astype(np.float32)
    coeff = rng.uniform(0.1, 3.0, size=(frames * C_, 16)).astype(np.float32)
    coeff.sort(axis=1)
    d_amp, d_coeff = torch.from_numpy(amp).to(dev), torch.from_numpy(coeff).to(dev)
    posts = torch.zeros((frames * C_, 64), dtype=torch.int16, device=dev)
    counts = torch.ones(frames * C_, dtype=torch.uint8, device=dev)
    dec = Decoder(ctx, C_, 256, 2048, floors=[f0], mappings=[{"coupling": [], "channel_floor": [0, 0]}])
    samples = (frames - 1) * 1024
    cap = samples + 2048
    out = torch.empty(C_ * cap, device=dev)
    import ctypes as C

    def step_b():
        dec.reset(-1)
        dec.ctx._check(capi.lib().vpz_decoder_set_floor0_data(dec._h, C.c_void_p(d_amp.data_ptr()), C.c_void_p(d_coeff.data_ptr()), 16))
        return dec.synth_raw(pk, res, posts, counts, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)

    dt, w = timed(ctx, step_b)
    byt = 4 * res.numel() + 4 * samples * C_
    print("(b) stereo, type-0 floor of order 16, %d frames (curve per bark index + the stereo fast path; VPZ_NO_F0_FUSED=1: planar temp + Floor0.Apply + one-channel kernel): %.3f ms/call  %.1f Msamples/s  "
          "%.0f GB/s algorithmic = %.3f of 8 TB/s" % (frames, dt * 1e3, samples * C_ / dt / 1e6, byt / dt / 1e9, byt / dt / 8e12), flush=True)
    dec.close()
    del res, out

    # (c) block sizes of the three-pass path
    for size0, size1 in ((512, 4096), (1024, 8192)):
        frames = 65536 * 2048 // size1
        flags = helpers.markov_block_flags(frames, seed=3)
        halves = np.where(flags & 1, size1 // 2, size0 // 2).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(halves * 2)])
        pk = make_packets(frames)
        pk["flags"] = flags | capi.PKT_NO_FLOOR
        pk["granule"] = -1
        pk["residue_offset"] = offs[:-1]
        res = torch.randn(int(offs[-1]), device=dev) * 2.0 ** -8
        dec = Decoder(ctx, 2, size0, size1)
        cap = int(halves.sum()) + 2 * size1
        out = torch.empty(2 * cap, device=dev)

        def step_c():
            dec.reset(-1)
            return dec.synth_raw(pk, res, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)

        dt, w = timed(ctx, step_c)
        smp = int(w[0]) * 2
        byt = 4 * int(offs[-1]) + 4 * smp
        print("(c) block sizes %d/%d, stereo, %d frames (synth_big_kernel; VPZ_NO_BIG=1: three passes over HBM): %.3f ms/call  %.1f Msamples/s  %.0f GB/s algorithmic = "
              "%.3f of 8 TB/s" % (size0, size1, frames, dt * 1e3, smp / dt / 1e6, byt / dt / 1e9, byt / dt / 8e12), flush=True)
        dec.close()
        del res, out
    ctx.close()


if __name__ == "__main__":
    main()
