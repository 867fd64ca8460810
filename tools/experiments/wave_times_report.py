#!/usr/bin/env python3
"""Reads a VPZ_STAMPS_DUMP file of a -DVPZ_WAVE_TIMES build: when each wave ran, on which CU / SIMD / slot."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
a = np.array([[int(r[k]) for k in ("run", "stream", "frames", "passes", "long_frames", "c0", "c1")] for r in rows], dtype=np.int64)
hw = a[:, 4]
t0, t1 = a[:, 5], a[:, 6]
base = t0.min()
t0 = t0 - base; t1 = t1 - base
dur = t1 - t0
slot = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("waves %d; kernel span %d ticks; start: min %d max %d; end: median %d max %d; mean duration %.0f" % (len(a), t1.max(), t0.min(), t0.max(), np.median(t1), t1.max(), dur.mean()))
print("residency: mean duration / span = %.3f" % (dur.mean() / t1.max()))
for s in sorted(set(slot)):
    m = slot == s
    print("wave slot %d: %4d waves, start %8.0f, end %8.0f, ticks per pass %.0f" % (s, m.sum(), t0[m].mean(), t1[m].mean(), (dur[m] / np.maximum(1, a[m, 3])).mean()))
half = a[:, 0] >= len(a) // 2
for nm, m in (("first half of the grid", ~half), ("second half", half)):
    print("%s: start %8.0f end %8.0f per pass %.0f; slots %s" % (nm, t0[m].mean(), t1[m].mean(), (dur[m] / np.maximum(1, a[m, 3])).mean(), np.bincount(slot[m])[:4]))
key = se * 1000 + sh * 100 + cu
print("distinct CUs (se, sh, cu):", len(set(key)), "waves per CU min/max", np.bincount(np.unique(key, return_inverse=True)[1]).min(), np.bincount(np.unique(key, return_inverse=True)[1]).max())
q = np.percentile(t1, [5, 25, 50, 75, 95, 100])
print("end-time percentiles 5/25/50/75/95/100:", q.astype(int))
