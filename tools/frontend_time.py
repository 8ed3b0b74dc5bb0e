import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_adata
for rows, cols, k, P in ((50, 100, 10, 1000), (250, 400, 20, 10000), (1000, 1000, 30, 10000)):
    ad = hex_adata(rows, cols, k, n_genes=0)
    sq.gr.nhood_enrichment(ad, "cluster", n_perms=16, seed=0)
    for _ in range(2):
        t = time.perf_counter(); sq.gr.nhood_enrichment(ad, "cluster", n_perms=P, seed=0); dt = time.perf_counter() - t
    print(f"nhood_enrichment front-end n={rows*cols} K={k} P={P}: {dt*1e3:.1f} ms -> {P/dt:.0f} perms/s", flush=True)
ad = hex_adata(100, 200, 10)
for name, fn in (("co_occurrence", lambda: sq.gr.co_occurrence(ad, "cluster")), ("ripley L", lambda: sq.gr.ripley(ad, "cluster", mode="L")),
                 ("ripley F", lambda: sq.gr.ripley(ad, "cluster", mode="F")), ("ripley G", lambda: sq.gr.ripley(ad, "cluster", mode="G")),
                 ("grid graph", lambda: sq.gr.spatial_neighbors_grid(ad))):
    fn(); t = time.perf_counter(); fn(); print(f"{name} n=20000: {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
