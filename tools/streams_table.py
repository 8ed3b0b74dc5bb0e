"""End-to-end wall time of DEFAULT front-end calls (numpy's streams: Squidpy's numbers) and of the throughput mode (rng="philox")
on BASELINE configs 1, 2, 5 (nhood_enrichment) and 3 (spatial_autocorr, 2000 of the 20 000 genes resident as float32 CSR) — the
table README / DESIGN quote.  GPU box:  python tools/streams_table.py > gpurun_out/streams_table.json"""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd

import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid, hex_grid_graph, knn_directed_graph

out = {}


def timed(fn, reps=3):
    fn()  # warm: lists, workspaces, module load
    best = np.inf
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


for name, rows, cols, k, P in (("config1", 50, 100, 10, 1000), ("config2", 250, 400, 20, 10_000), ("config5_default_n_perms", 1000, 1000, 30, 1000),
                               ("config5", 1000, 1000, 30, 100_000)):
    n = rows * cols
    labels = np.random.default_rng(0).integers(0, k, n)
    adata = sq.AnnDataLite(obs=pd.DataFrame({"cluster": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(k)])}),
                           obsp={"spatial_connectivities": hex_grid_graph(rows, cols)})
    rec = {"spots": n, "clusters": k, "n_perms": P}
    for rng in ("numpy", "philox"):
        dt = timed(lambda: sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=1, copy=True, rng=rng, show_progress_bar=False), reps=2 if P > 10_000 else 3)
        rec[rng] = {"seconds": dt, "perms_per_s": P / dt}
    out[name] = rec
    print(name, json.dumps(rec), file=sys.stderr, flush=True)

rows, cols, G, P = 250, 400, 20_000, 1000
n = rows * cols
from scipy import sparse

xy = hex_grid(rows, cols) + np.random.default_rng(5).normal(0.0, 1.0, (n, 2))
X = sparse.random(n, G, density=0.1, format="csr", dtype=np.float32, random_state=3, data_rvs=lambda s: np.random.default_rng(4).gamma(2.0, 1.0, s).astype(np.float32))
adata = sq.AnnDataLite(X=X, obs=pd.DataFrame(index=[f"s{i}" for i in range(n)]), var=pd.DataFrame(index=[f"g{i}" for i in range(G)]),
                       obsp={"spatial_connectivities": knn_directed_graph(xy, 6)})
rec = {"spots": n, "genes": G, "n_perms": P, "input": "CSR float32, 10 % density, directed kNN-6 graph"}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for mode in ("moran", "geary"):
        for rng in ("numpy", "philox"):
            dt = timed(lambda: sq.gr.spatial_autocorr(adata, mode=mode, genes=list(adata.var_names), n_perms=P, seed=1, copy=True, rng=rng, show_progress_bar=False), reps=2)
            rec[f"{mode}_{rng}"] = {"seconds": dt, "genes_per_s": G / dt}
out["config3"] = rec
print("config3", json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
