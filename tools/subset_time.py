"""Developer tool (GPU box): `spatial_autocorr` on a gene SUBSET (the reference's default: the highly variable genes) of a
sparse float32 expression matrix, config-3 shape (1e5 cells x 20 000 genes, 10 % density), 2000 genes asked for: the host
subsetting the reference does (`adata[:, genes].X`) against selecting the columns on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, scipy.sparse as sp
import squidpy_amd as sq
from squidpy_amd import _lib as L
from squidpy_amd.gr import _ppatterns as pp
from squidpy_amd._synthetic import hex_grid_graph

n, G, H = 100_000, int(os.environ.get("G", "20000")), int(os.environ.get("H", "2000"))
rng = np.random.default_rng(7)
parts = []
for r0 in range(0, n, 4000):
    blk = sp.csr_matrix(rng.random((4000, G)) < 0.1, dtype=np.float32)
    blk.data = rng.integers(1, 30, blk.nnz).astype(np.float32)
    parts.append(blk)
Xs = sp.vstack(parts, format="csr"); del parts
hv = np.zeros(G, dtype=bool); hv[rng.choice(G, H, replace=False)] = True
var = pd.DataFrame({"highly_variable": hv}, index=[f"g{i}" for i in range(G)])
obs = pd.DataFrame(index=[f"s{i}" for i in range(n)])
graph = hex_grid_graph(250, 400)
ctx = L.default_context()
for fmt in ("csr", "csc", "dense32"):
    X = Xs if fmt == "csr" else Xs.tocsc() if fmt == "csc" else Xs[:, :4000].toarray()
    v = var if fmt != "dense32" else pd.DataFrame({"highly_variable": hv[:4000] | (np.arange(4000) % 3 == 0)}, index=var.index[:4000])
    adata = sq.AnnDataLite(X=X, obs=obs, var=v, obsp={"spatial_connectivities": graph})
    sq.gr.spatial_autocorr(adata, genes=list(adata.var_names[:64]), n_perms=16, seed=1, copy=True)  # warm
    res = {}
    for where in ("device", "host"):
        real = pp._ColumnSelection.worthwhile
        if where == "host":
            pp._ColumnSelection.worthwhile = staticmethod(lambda base, cols: False)
        ctx.timer_enable(True); ctx.timer_reset()
        t0 = time.perf_counter()
        res[where] = sq.gr.spatial_autocorr(adata, mode="moran", n_perms=1000, seed=1, copy=True)
        dt = time.perf_counter() - t0
        rep = ctx.timer_report(); ctx.timer_enable(False)
        pp._ColumnSelection.worthwhile = real
        top = sorted(rep.items(), key=lambda kv: -kv[1][1])[:5]
        print(fmt, where, f"{len(res[where])} genes, end to end {dt:.3f} s; kernels:", {k: (v[0], round(v[1], 1)) for k, v in top}, flush=True)
    pd.testing.assert_frame_equal(res["device"], res["host"], check_exact=True)
    print(fmt, "frames identical", flush=True)
