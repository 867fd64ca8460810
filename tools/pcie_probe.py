#!/usr/bin/env python3
"""What the host link gives on this box: pinned host <-> device copies of the sizes the end-to-end leg moves (one sub-batch of
16 real stereo streams: ~54 MB of residue in, ~54 MB of float PCM out), alone, both directions at once, and on two streams."""
import time
import torch

dev = torch.device("cuda", 0)
for mb in (8, 54, 428):
    n = mb * (1 << 20) // 4
    h_in = torch.empty(n, dtype=torch.float32, pin_memory=True).normal_()
    h_out = torch.empty(n, dtype=torch.float32, pin_memory=True)
    d_in = torch.empty(n, dtype=torch.float32, device=dev)
    d_out = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t_h2d = timed(lambda: d_in.copy_(h_in, non_blocking=True))
    t_d2h = timed(lambda: h_out.copy_(d_out, non_blocking=True))

    def both():
        with torch.cuda.stream(s1):
            d_in.copy_(h_in, non_blocking=True)
        with torch.cuda.stream(s2):
            h_out.copy_(d_out, non_blocking=True)
    t_both = timed(both)
    gb = n * 4 / 1e9
    print("%4d MB: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s each way (%.2f ms)" % (mb, gb / t_h2d, gb / t_d2h, gb / t_both, t_both * 1e3), flush=True)
