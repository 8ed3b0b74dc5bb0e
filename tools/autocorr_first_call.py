"""Developer tool (GPU box): spatial_autocorr end to end at config 3's size in the order moran -> geary, with a cProfile of
the second call (a multi-second stall showed up there once: is it host code, the allocator, or the driver?)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid, hex_grid_graph
rows, cols, G, P = 250, 400, int(os.environ.get("G", 8192)), 1000
n = rows * cols
rng = np.random.default_rng(1)
adj = hex_grid_graph(rows, cols)
X = rng.gamma(2.0, 1.0, size=(n, G))
adata = sq.AnnDataLite(X=X, obs=pd.DataFrame(index=[str(i) for i in range(n)]), obsm={"spatial": hex_grid(rows, cols)}, obsp={"spatial_connectivities": adj})
sq.gr.spatial_autocorr(adata, mode="moran", genes=list(adata.var_names[:64]), n_perms=10, seed=0, copy=True)
for mode in os.environ.get("ORDER", "moran,geary,geary").split(","):
    pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
    sq.gr.spatial_autocorr(adata, mode=mode, genes=list(adata.var_names), n_perms=P, seed=0, copy=True)
    pr.disable(); dt = time.perf_counter() - t
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(6)
    print(f"== {mode}: {dt:.3f} s"); print("\n".join(l for l in s.getvalue().splitlines()[6:16]), flush=True)
