import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid_graph
for side, K, P in ((316, 20, 10_000), (1000, 30, 100_000)):
    adj = hex_grid_graph(side, side); n = adj.shape[0]
    obs = pd.DataFrame({"cluster": pd.Categorical(np.random.default_rng(0).integers(0, K, n).astype(str))})
    adata = sq.AnnDataLite(obs=obs, obsp={"spatial_connectivities": adj})
    sq.gr.nhood_enrichment(adata, "cluster", n_perms=64, seed=0, copy=True)
    for rng in ("philox", "numpy"):
        t = time.perf_counter(); r = sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=0, copy=True, rng=rng); dt = time.perf_counter() - t
        print(f"nhood_enrichment n={n} K={K} P={P} rng={rng}: {dt:.3f} s -> {P/dt:.0f} perms/s", flush=True)
