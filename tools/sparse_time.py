"""Developer tool (GPU box): where the time of `spatial_autocorr` on a sparse float32 expression matrix goes (config-3 shape,
10 % density): upload, per-block expansion, statistics — HIP-event kernel times and host wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, scipy.sparse as sp
import squidpy_amd as sq
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph

n, G = 100_000, int(os.environ.get("G", "20000"))
rng = np.random.default_rng(7)
parts = []
for r0 in range(0, n, 4000):
    mask = rng.random((4000, G)) < 0.1
    blk = sp.csr_matrix(mask, dtype=np.float32)
    blk.data = rng.integers(1, 30, blk.nnz).astype(np.float32)
    parts.append(blk)
Xs = sp.vstack(parts, format="csr"); del parts
print("nnz", Xs.nnz, "index dtype", Xs.indices.dtype, flush=True)
adata = sq.AnnDataLite(X=Xs, obs=pd.DataFrame(index=[f"s{i}" for i in range(n)]), obsp={"spatial_connectivities": hex_grid_graph(250, 400)})
ctx = L.default_context()
sq.gr.spatial_autocorr(adata, genes=list(adata.var_names[:256]), n_perms=64, seed=1, copy=True)
for fmt in ("csr", "csc"):
    a = adata if fmt == "csr" else sq.AnnDataLite(X=Xs.tocsc(), obs=adata.obs, obsp=adata.obsp)
    t0 = time.perf_counter(); dm = L.DeviceMatrix(ctx, a.X); ctx.sync(); t_up = time.perf_counter() - t0; dm.close()
    ctx.timer_enable(True); ctx.timer_reset()
    t0 = time.perf_counter()
    df = sq.gr.spatial_autocorr(a, mode="moran", n_perms=1000, seed=1, copy=True)
    dt = time.perf_counter() - t0
    rep = ctx.timer_report(); ctx.timer_enable(False)
    top = sorted(rep.items(), key=lambda kv: -kv[1][1])[:8]
    print(fmt, f"end to end {dt:.3f} s; DeviceMatrix upload alone {t_up:.3f} s; kernels:", {k: (v[0], round(v[1], 1)) for k, v in top}, "sum ms", round(sum(v[1] for v in rep.values()), 1), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); sq.gr.spatial_autocorr(adata, mode="moran", n_perms=1000, seed=1, copy=True); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
