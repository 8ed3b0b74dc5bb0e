"""Developer tool (GPU box): the two permutation kernels of spatial_autocorr on the config-3 shape, per-kernel HIP-event times.

    python tools/autocorr_kernels_time.py [G] [P]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.preprocessing import normalize

from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rows, cols = 250, 400
n = rows * cols
ctx = L.default_context()
g = normalize(hex_grid_graph(rows, cols), norm="l1", axis=1)
vals = np.random.default_rng(1).gamma(2.0, 1.0, size=(G, n))
graph = L.Graph(ctx, g, with_data=True)
plan = L.AutocorrPlan(ctx, graph, vals)
ref = {}
for kern in ("gather", "lds"):
    os.environ["SQGR_AUTOCORR_KERNEL"] = kern
    for mode in ("moran", "geary"):
        plan.perms(mode, seed=1, perm_begin=0, perm_end=64)  # warm
        ctx.timer_enable(True)
        ctx.timer_reset()
        t = time.perf_counter()
        sims = plan.perms(mode, seed=7, perm_begin=0, perm_end=P)
        dt = time.perf_counter() - t
        rep = ctx.timer_report()
        ctx.timer_enable(False)
        ks = {k: round(v[1], 3) for k, v in rep.items() if k.startswith("autocorr")}
        evals = G * P * n
        dot_ms = sum(v for k, v in ks.items() if "perm_dot" in k)
        print(f"{kern:6s} {mode:5s}: {dt*1e3:8.1f} ms wall -> {G/dt:8.0f} genes/s; dot {dot_ms:.1f} ms = {evals/dot_ms/1e9:.2f}e12 evals/s; kernels_ms={ks}", flush=True)
        if mode in ref:
            d = np.abs(sims - ref[mode])
            print(f"       max |lds - gather| = {np.nanmax(d):.3e} (values ~ {np.nanstd(sims):.2e})", flush=True)
        else:
            ref[mode] = sims
