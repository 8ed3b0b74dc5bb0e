import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from squidpy_amd import _lib as L
ctx = L.default_context()
rng = np.random.default_rng(0)
for n, k in ((33000, 2), (33000, 1), (100000, 2)):
    pts = rng.random((n, 2)) * 1000
    L.knn_dist(ctx, pts[:100], pts, k)
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter(); d = L.knn_dist(ctx, pts, pts, k); dt = time.perf_counter() - t
    print(n, k, "wall", round(dt*1e3, 2), "ms", {a: b for a, b in ctx.timer_report().items() if b[0]}, flush=True)
    ctx.timer_enable(False)
# ripley G at 1e6 points: which kernels, how long
import pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid
n = 1000 * 1000
xy = hex_grid(1000, 1000) + rng.normal(0, 5, (n, 2))
adata = sq.AnnDataLite(obs=pd.DataFrame({"cluster": pd.Categorical(rng.integers(0, 30, n).astype(str))}), obsm={"spatial": xy})
sq.gr.ripley(adata, "cluster", mode="G", copy=True, seed=0, n_simulations=2)
ctx.timer_enable(True); ctx.timer_reset()
t = time.perf_counter(); sq.gr.ripley(adata, "cluster", mode="G", copy=True, seed=0); dt = time.perf_counter() - t
print("ripley G wall", round(dt, 3), {a: (b[0], round(b[1], 1)) for a, b in ctx.timer_report().items() if b[0]})
