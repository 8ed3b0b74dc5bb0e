"""ligrec front-end (pandas host logic + device permutations) timing with a cProfile summary of the host part."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, scipy.sparse as sp
import squidpy_amd as sq

n, g, k, n_inter, P = 100_000, 2000, 25, 1000, 1000
rng = np.random.default_rng(0)
x = sp.random(n, g, density=0.1, format="csr", random_state=rng, data_rvs=lambda s: rng.gamma(2.0, 1.0, s))
obs = pd.DataFrame({"cluster": pd.Categorical(rng.integers(0, k, n).astype(str))})
adata = sq.AnnDataLite(X=x, obs=obs, var=pd.DataFrame(index=[f"G{i}" for i in range(g)]))
pairs = pd.DataFrame({"source": [f"G{i}" for i in rng.integers(0, 400, n_inter)], "target": [f"G{i}" for i in rng.integers(0, 400, n_inter)]})
sq.gr.ligrec(adata, "cluster", interactions=pairs.iloc[:10], n_perms=10, use_raw=False, copy=True, seed=0)
for r in ("philox", "numpy"):
    t = time.perf_counter()
    res = sq.gr.ligrec(adata, "cluster", interactions=pairs, n_perms=P, use_raw=False, copy=True, seed=0, rng=r)
    print(f"ligrec rng={r}: {n} cells, {g} genes in adata, {len(res['means'])} interactions x {res['means'].shape[1]} cluster pairs, {P} perms: {time.perf_counter() - t:.3f} s", flush=True)
pr = cProfile.Profile(); pr.enable()
sq.gr.ligrec(adata, "cluster", interactions=pairs, n_perms=P, use_raw=False, copy=True, seed=0)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
