import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid
ctx = L.default_context()
for rows, cols in ((100, 200), (300, 400), (500, 1000)):
    n = rows * cols
    rng = np.random.default_rng(0)
    xy = hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
    labs = rng.integers(0, 30, n).astype(np.int32)
    d = np.hypot(xy[:, 0].max(), xy[:, 1].max()) / 2
    thr = np.linspace(100, d, 49, dtype=np.float32) ** 2
    L.cooccur_counts(ctx, xy[:100, 0], xy[:100, 1], labs[:100], 30, thr)
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter(); c = L.cooccur_counts(ctx, xy[:, 0], xy[:, 1], labs, 30, thr); dt = time.perf_counter() - t
    ms, cnt = ctx.timer_get("cooccur")
    print(f"n={n}: wall {dt:.3f}s kernel {ms:.1f} ms -> {n*(n-1)/ (ms*1e-3):.3e} ordered pairs/s", flush=True)
