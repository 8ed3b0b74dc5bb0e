"""Developer tool (GPU box): Moran's I at the config-3 shape with few permutations — the LDS kernel on virtual permutations
(lds-split) against the gather kernel, and the LDS kernel proper at 1000 permutations for reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.preprocessing import normalize
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
ctx = L.default_context()
rows, cols, G = 250, 400, 2048
n = rows * cols
g = normalize(hex_grid_graph(rows, cols), norm="l1", axis=1)
vals = np.random.default_rng(11).gamma(2.0, 1.0, size=(G, n))
graph = L.Graph(ctx, g, with_data=True)
plan = L.AutocorrPlan(ctx, graph, vals)
res = {}
for mode in ("moran", "geary"):
    for P in (16, 50, 100, 256):
        for kern in ("lds-split", "gather"):
            os.environ["SQGR_AUTOCORR_KERNEL"] = kern
            plan.perms(mode, seed=1, perm_begin=0, perm_end=P)
            ctx.sync(); ctx.timer_enable(True); ctx.timer_reset()
            t = time.perf_counter(); sims = plan.perms(mode, seed=2, perm_begin=0, perm_end=P); ctx.sync(); dt = time.perf_counter() - t
            rep = {k: round(v[1], 2) for k, v in ctx.timer_report().items() if v[0]}
            ctx.timer_enable(False)
            res[(mode, P, kern)] = sims
            print(f"{mode} P={P} {kern}: {G/dt:.0f} genes/s wall {dt*1e3:.1f} ms kernels_ms={rep}", flush=True)
        a, b = res[(mode, P, "lds-split")], res[(mode, P, "gather")]
        print("   max rel diff split vs gather:", float(np.nanmax(np.abs(a - b) / (np.abs(b) + 1e-12))), flush=True)
