"""Developer tool (GPU box): cProfile of a DEFAULT nhood_enrichment call at 1e6 spots (second call: caches warm)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid_graph
rows = cols = 1000
n, k, P = rows * cols, 30, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
labels = np.random.default_rng(0).integers(0, k, n)
adata = sq.AnnDataLite(obs=pd.DataFrame({"cluster": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(k)])}),
                       obsp={"spatial_connectivities": hex_grid_graph(rows, cols)})
f = lambda: sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=1, copy=True, show_progress_bar=False)
f(); f()
t = time.perf_counter(); f(); print("call ms", (time.perf_counter() - t) * 1e3)
pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
