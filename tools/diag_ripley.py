import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.neighbors import KDTree
from oracle import restate as O
from squidpy_amd import _lib as L
ctx = L.default_context()
m = 3000
rng = np.random.default_rng(m)
pts = np.round(rng.random((m, 2)) * 40, 1)
support = np.linspace(0, 25, 50)
kd = KDTree(pts).two_point_correlation(pts, support, dualtree=True) - m
bf = O.pair_counts_bruteforce(pts, support)
gpu = L.pair_counts(ctx, pts, support)
print("kd-bf", np.nonzero(kd - bf)[0], "gpu-bf", np.nonzero(gpu - bf)[0], (gpu - bf)[np.nonzero(gpu - bf)[0]])
thr = L.sqrt_thresholds(support)
d = pts[:, None, :] - pts[None, :, :]
d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
for i in np.nonzero(gpu - bf)[0]:
    t = thr[i]
    near = np.argwhere(np.abs(d2 - t) <= 4 * np.spacing(t))
    print("idx", i, "r", repr(support[i]), "thr", repr(t), "near pairs", len(near))
    for a, b in near[:6]:
        print("   ", a, b, repr(d2[a, b]), d2[a, b] <= t, repr(pts[a]), repr(pts[b]))
# direct single-pair checks through the kernel
for a, b in near[:3]:
    sub = np.stack([pts[a], pts[b]])
    print("pair alone:", L.pair_counts(ctx, sub, support)[i], "expected", 2 * int(d2[a, b] <= t))
