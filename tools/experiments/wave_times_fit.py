#!/usr/bin/env python3
"""Least-squares fit of the waves' durations (-DVPZ_WAVE_TIMES dump) to their runs' composition: what a long pass, a short
block alone, a batch pass and a block riding in a batch cost -- the weights the host's run cutting should use."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
a = np.array([[int(r[k]) for k in ("c0", "c1", "c2", "c3", "c4", "c5", "c6", "frames")] for r in rows], dtype=np.int64)
ok = a[:, 1] > a[:, 0]
a = a[ok]
dur = (a[:, 1] - a[:, 0]).astype(float)
X = np.stack([np.ones(len(a)), a[:, 2], a[:, 3], a[:, 4], a[:, 5]], 1).astype(float)
coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
print("waves %d, duration mean %.0f max %.0f (max / mean %.3f), std %.0f" % (len(a), dur.mean(), dur.max(), dur.max() / dur.mean(), dur.std()))
print("fit: fixed %.0f, long pass %.0f, short alone %.0f, batch pass %.0f, block in a batch %.0f; residual std %.0f" % (*coef, (dur - X @ coef).std()))
L = coef[1]
print("in eighths of a long pass: short alone %.2f, batch pass + 1st block %.2f, every further block %.2f, fixed per run %.2f" %
      (8 * coef[2] / L, 8 * (coef[3] + coef[4]) / L, 8 * coef[4] / L, 8 * coef[0] / L))
model = 8 * a[:, 2] + 6 * a[:, 3] + 6 * a[:, 4] + 3 * (a[:, 5] - a[:, 4])
print("host model units: mean %.1f max %d min %d; corr(model, duration) %.3f" % (model.mean(), model.max(), model.min(), np.corrcoef(model, dur)[0, 1]))
pre = a[:, 6]
for k in sorted(set(pre)):
    m = pre == k
    print("pre_kind %d: %d waves, mean duration %.0f, mean residual %.0f" % (k, m.sum(), dur[m].mean(), (dur - X @ coef)[m].mean()))
