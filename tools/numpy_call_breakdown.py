"""Developer tool (GPU box): where a DEFAULT nhood_enrichment call (numpy's streams) spends its time — host seeding, kernels."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd import _lib
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd._utils import pcg64_states

ctx = _lib.default_context()
for rows, cols, k, P in ((50, 100, 10, 1000), (250, 400, 20, 1000), (250, 400, 20, 10000), (1000, 1000, 30, 1000), (1000, 1000, 30, 8192)):
    n = rows * cols
    labels = np.random.default_rng(0).integers(0, k, n)
    adata = sq.AnnDataLite(obs=pd.DataFrame({"cluster": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(k)])}),
                           obsp={"spatial_connectivities": hex_grid_graph(rows, cols)})
    f = lambda: sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=1, copy=True, show_progress_bar=False)
    f(); f()
    t0 = time.perf_counter(); pcg64_states(1, P); t_seed = time.perf_counter() - t0
    ctx.timer_enable(True); ctx.timer_reset()
    t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
    rep = ctx.timer_report(); ctx.timer_enable(False)
    ks = {kk: round(v[1], 3) for kk, v in rep.items() if v[0] > 0}
    print(json.dumps({"spots": n, "K": k, "n_perms": P, "call_ms": round(dt * 1e3, 2), "seed_states_ms": round(t_seed * 1e3, 2), "kernel_ms_sum": round(sum(ks.values()), 2), "kernels_ms": ks}), flush=True)
