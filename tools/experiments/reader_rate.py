"""Tuning helper: one file through the VorbisReader mirror (ReadSamples loop) for several batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vorbispizza_amd import Context, capi
from vorbispizza_amd.front import VorbisReader
ctx = Context(0)
for name in ("3test.ogg", "issue6test.ogg", "2test.ogg"):
    data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    for batch in (32, 128, 1024):
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            r = VorbisReader(ctx, data, batch_packets=batch)
            C = r.Channels
            buf = np.zeros(C * 4096, dtype=np.float32)
            tot = 0
            mismatches = 0
            while True:
                try:
                    n = r.ReadSamples(buf)
                except capi.SynthError as e:  # issue6test.ogg's trailing packet: that one Read throws (StreamDecoder.cs:777-778)
                    mismatches += 1
                    if e.status != capi.E_WINDOW_MISMATCH or mismatches > 4:
                        raise  # (anything else -- or a failure that persists -- must not spin here holding the GPU)
                    continue
                if n == 0: break
                tot += n
            dt = time.perf_counter() - t0
            r.Dispose()
            best = min(best, dt)
        print("%s batch %4d: %d samples x %d ch in %.2f ms = %.1f Msamples/s" % (name, batch, tot, C, best*1e3, tot*C/best/1e6))
ctx.close()
