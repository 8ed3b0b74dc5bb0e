"""Front-end wall time at C4 scale (1e6 points, 30 clusters) with a cProfile summary: host overheads show up here."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid

rows = cols = int(os.environ.get("SIDE", 1000))
n = rows * cols
rng = np.random.default_rng(0)
xy = hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
obs = pd.DataFrame({"cluster": pd.Categorical(rng.integers(0, 30, n).astype(str))})
adata = sq.AnnDataLite(obs=obs, obsm={"spatial": xy})
small = sq.AnnDataLite(obs=obs.iloc[:2000].copy(), obsm={"spatial": xy[:2000]})
sq.gr.co_occurrence(small, "cluster", copy=True); sq.gr.ripley(small, "cluster", mode="L", copy=True, n_simulations=2, n_observations=50)

def prof(label, fn):
    pr = cProfile.Profile(); t = time.perf_counter(); pr.enable(); fn(); pr.disable(); dt = time.perf_counter() - t
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8)
    print(f"== {label}: {dt:.3f} s"); print("\n".join(l for l in s.getvalue().splitlines()[6:18]), flush=True)

prof(f"co_occurrence n={n}", lambda: sq.gr.co_occurrence(adata, "cluster", interval=50, copy=True))
for mode in ("L", "F", "G"):
    prof(f"ripley {mode} n={n}", lambda: sq.gr.ripley(adata, "cluster", mode=mode, copy=True, seed=0))
