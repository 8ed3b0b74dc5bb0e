"""Developer tool (GPU box): nhood_enrichment(rng="philox") through the front end on 1e6 observations in RANDOM order — with the
plan-internal renumbering (default) and without (SQGR_NHOOD_RENUMBER=0): first and second call, and what the twin costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, scipy.sparse as sp
import squidpy_amd as sq
from squidpy_amd import _lib
from squidpy_amd._synthetic import hex_grid, hex_grid_graph
rows = cols = 1000; n = rows * cols; k = 30
rng = np.random.default_rng(0)
adj = hex_grid_graph(rows, cols).tocoo()
perm = rng.permutation(n); inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
shuf = sp.csr_matrix((adj.data, (inv[adj.row], inv[adj.col])), shape=(n, n)); shuf.sort_indices()
lab = rng.integers(0, k, n)
adata = sq.AnnDataLite(obs=pd.DataFrame({"cluster": pd.Categorical.from_codes(lab, [f"c{i}" for i in range(k)])}),
                       obsm={"spatial": hex_grid(rows, cols)[perm]}, obsp={"spatial_connectivities": shuf})
ctx = _lib.default_context()
res = {}
for P in (1000, 10000, 100000):
    for mode in ("0", "auto"):
        os.environ["SQGR_NHOOD_RENUMBER"] = mode
        _lib.clear_graph_cache()
        ts = []
        for rep in range(3):
            t = time.perf_counter(); r = sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=1, copy=True, rng="philox", show_progress_bar=False); ts.append(time.perf_counter() - t)
        res[(P, mode)] = r.zscore
        print(f"n_perms={P} renumber={mode}: calls {[round(x * 1e3, 1) for x in ts]} ms -> {P / ts[-1]:.0f} perms/s", flush=True)
    print("  z-scores equal:", bool(np.array_equal(res[(P, '0')], res[(P, 'auto')])))
ctx.timer_enable(True); ctx.timer_reset()
sq.gr.nhood_enrichment(adata, "cluster", n_perms=1000, seed=1, copy=True, rng="philox", show_progress_bar=False)
print({k_: round(v[1], 2) for k_, v in ctx.timer_report().items() if v[0] and ("graph" in k_ or "order" in k_)})
import cProfile, pstats
os.environ["SQGR_NHOOD_RENUMBER"] = "auto"
pr = cProfile.Profile(); pr.enable()
sq.gr.nhood_enrichment(adata, "cluster", n_perms=1000, seed=1, copy=True, rng="philox", show_progress_bar=False)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
