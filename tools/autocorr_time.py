"""spatial_autocorr front-end timing: device generator vs numpy streams (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid, hex_grid_graph

rows, cols, G, P = 250, 400, int(os.environ.get("G", 4096)), 1000
n = rows * cols
rng = np.random.default_rng(1)
adj = hex_grid_graph(rows, cols)
X = rng.gamma(2.0, 1.0, size=(n, G))
adata = sq.AnnDataLite(X=X, obs=pd.DataFrame(index=[str(i) for i in range(n)]), obsm={"spatial": hex_grid(rows, cols)},
                       obsp={"spatial_connectivities": adj})
sq.gr.spatial_autocorr(adata, mode="moran", genes=list(adata.var_names[:64]), n_perms=10, seed=0, copy=True)
for mode in ("moran", "geary"):
    for r in ("philox", "numpy"):
        t = time.perf_counter()
        sq.gr.spatial_autocorr(adata, mode=mode, genes=list(adata.var_names), n_perms=P, seed=0, copy=True, rng=r)
        dt = time.perf_counter() - t
        print(f"{mode} rng={r}: n={n} G={G} P={P}: {dt:.3f} s -> {G / dt:.0f} genes/s", flush=True)
