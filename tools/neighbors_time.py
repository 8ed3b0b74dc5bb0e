"""spatial_neighbors front-ends at 1e6 spots with a cProfile summary (run on the GPU box)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid
n_side = int(os.environ.get("SIDE", 1000))
xy = hex_grid(n_side, n_side)
n = len(xy)
adata = sq.AnnDataLite(obs=pd.DataFrame(index=[str(i) for i in range(n)]), obsm={"spatial": xy})
sq.gr.spatial_neighbors_grid(sq.AnnDataLite(obs=pd.DataFrame(index=[str(i) for i in range(100)]), obsm={"spatial": xy[:100]}), copy=True)
for label, fn in (("grid n_neighs=6", lambda: sq.gr.spatial_neighbors_grid(adata, copy=True)),
                  ("knn k=6", lambda: sq.gr.spatial_neighbors_knn(adata, n_neighs=6, copy=True)),
                  ("radius 150", lambda: sq.gr.spatial_neighbors_radius(adata, radius=150.0, copy=True)),
                  ("grid 2 rings", lambda: sq.gr.spatial_neighbors_grid(adata, n_rings=2, copy=True))):
    pr = cProfile.Profile(); t = time.perf_counter(); pr.enable(); fn(); pr.disable(); dt = time.perf_counter() - t
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(7)
    print(f"== {label} n={n}: {dt:.3f} s"); print("\n".join(s.getvalue().splitlines()[6:15]), flush=True)
