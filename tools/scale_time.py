"""Developer tool: time autocorr / co-occurrence / ripley at BASELINE config scale on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid, hex_grid_graph
from sklearn.preprocessing import normalize
which = sys.argv[1:] or ["autocorr", "cooccur", "ripley"]
ctx = L.default_context()
rng = np.random.default_rng(1)
if "autocorr" in which:
    rows, cols, G, P = 250, 400, int(os.environ.get("G", 2048)), 1000
    n = rows * cols
    g = normalize(hex_grid_graph(rows, cols), norm="l1", axis=1)
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    graph = L.Graph(ctx, g)
    t = time.perf_counter(); plan = L.AutocorrPlan(ctx, graph, vals); t_prep = time.perf_counter() - t
    for mode in ("moran", "geary"):
        plan.perms(mode, seed=1, perm_begin=0, perm_end=16)
        ctx.timer_enable(True); ctx.timer_reset()
        t = time.perf_counter(); s = plan.scores(mode); sims = plan.perms(mode, seed=1, perm_begin=0, perm_end=P); dt = time.perf_counter() - t
        rep = {k: (v[0], round(v[1], 2)) for k, v in ctx.timer_report().items() if v[0]}
        ctx.timer_enable(False)
        kms = sum(v[1] for k, v in rep.items() if "perm_dot" in k)
        print(f"autocorr {mode}: N={n} G={G} P={P}: wall {dt:.3f}s (prep {t_prep:.2f}s) -> {G/dt:.0f} genes/s; perm_dot {kms:.1f} ms -> "
              f"{G*(P)*n*8/ (kms*1e-3)/1e12:.2f} TB/s gather; kernels {rep}", flush=True)
    plan.close()
if "cooccur" in which:
    for rows, cols in ((1000, 1000),):
        n = rows * cols
        xy = hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
        labs = rng.integers(0, 30, n).astype(np.int32)
        d = np.hypot(xy[:, 0].max(), xy[:, 1].max()) / 2
        thr = np.linspace(100, d, 49, dtype=np.float32) ** 2
        ctx.timer_enable(True); ctx.timer_reset()
        t = time.perf_counter(); c = L.cooccur_counts(ctx, xy[:, 0], xy[:, 1], labs, 30, thr); dt = time.perf_counter() - t
        ms, cnt = ctx.timer_get("cooccur"); ctx.timer_enable(False)
        assert c[..., -1].sum() <= n * (n - 1)
        print(f"cooccur n={n}: wall {dt:.3f}s kernel {ms:.1f} ms -> {n*(n-1)/(ms*1e-3):.3e} ordered pairs/s", flush=True)
if "ripley" in which:
    rows = cols = 1000
    n = rows * cols
    xy = hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
    labs = rng.integers(0, 30, n)
    support = np.linspace(0, 60000, 50)
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter()
    tot = 0
    for c in range(30):
        pts = xy[labs == c]
        pc = L.pair_counts(ctx, pts, support); tot += len(pts) * (len(pts) - 1)
    dt = time.perf_counter() - t
    ms, cnt = ctx.timer_get("ripley_pair"); 
    print(f"ripley L pair counts, 30 clusters of ~{n//30}: wall {dt:.3f}s kernel {ms:.1f} ms -> {tot/(ms*1e-3):.3e} ordered pairs/s", flush=True)
    ctx.timer_reset()
    t = time.perf_counter(); dd = L.knn_dist(ctx, xy[labs != 0], xy[labs == 0], 2); dt = time.perf_counter() - t
    ms, cnt = ctx.timer_get("ripley_knn"); ctx.timer_enable(False)
    print(f"ripley G kNN (k=2) {len(dd)} queries x {int((labs==0).sum())} refs: wall {dt:.3f}s kernel {ms:.1f} ms -> {len(dd)*int((labs==0).sum())/(ms*1e-3):.3e} pairs/s", flush=True)
