"""cProfile of the spatial_autocorr front-end (where the non-kernel time goes)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid, hex_grid_graph
rows, cols, G, P = 250, 400, int(os.environ.get("G", 4096)), 1000
n = rows * cols
rng = np.random.default_rng(1)
X = rng.gamma(2.0, 1.0, size=(n, G))
adata = sq.AnnDataLite(X=X, obs=pd.DataFrame(index=[str(i) for i in range(n)]), obsm={"spatial": hex_grid(rows, cols)},
                       obsp={"spatial_connectivities": hex_grid_graph(rows, cols)})
sq.gr.spatial_autocorr(adata, mode="moran", genes=list(adata.var_names[:64]), n_perms=10, seed=0, copy=True)
ctx = sq._lib.default_context(); ctx.timer_enable(True); ctx.timer_reset()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
sq.gr.spatial_autocorr(adata, mode="moran", n_perms=P, seed=0, copy=True)
pr.disable(); dt = time.perf_counter() - t
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
print(f"spatial_autocorr moran n={n} G={G} P={P}: {dt:.3f} s; kernels {sum(v[1] for v in ctx.timer_report().values()):.0f} ms")
print("\n".join(s.getvalue().splitlines()[6:22]))
