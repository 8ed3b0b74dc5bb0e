"""Synthetic inputs of the benchmark configurations: hex-lattice point clouds with their
6-neighbour grid graph (what ``squidpy.gr.spatial_neighbors_grid`` produces for Visium-like data,
/root/reference/src/squidpy/gr/neighbors.py:335-419), uniform cluster labels, gamma expression."""

from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import sparse

from ._anndata_lite import AnnDataLite


def hex_grid(rows: int, cols: int, scale: float = 100.0) -> np.ndarray:
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    x = (c + 0.5 * (r % 2)).ravel() * scale
    y = (r * (np.sqrt(3.0) / 2.0)).ravel() * scale
    return np.stack([x, y], axis=1)


def hex_grid_graph(rows: int, cols: int) -> sparse.csr_matrix:
    """Closed-form CSR of the hex lattice's 6-neighbourhood (float32 ones, int32 indices, sorted rows)."""
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    r, c = r.ravel(), c.ravel()
    odd = r % 2
    cand = [(r, c - 1), (r, c + 1), (r - 1, c - 1 + odd), (r - 1, c + odd), (r + 1, c - 1 + odd), (r + 1, c + odd)]
    me = r * cols + c
    src, dst = [], []
    for rr, cc in cand:
        ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
        src.append(me[ok])
        dst.append((rr * cols + cc)[ok])
    src, dst = np.concatenate(src), np.concatenate(dst)
    n = rows * cols
    g = sparse.csr_matrix((np.ones(len(src), dtype=np.float32), (src, dst)), shape=(n, n))
    g.sort_indices()
    g.indices = g.indices.astype(np.int32)
    g.indptr = g.indptr.astype(np.int32)
    return g


def knn_directed_graph(xy: np.ndarray, k: int = 6, ctx=None) -> sparse.csr_matrix:
    """The directed k-nearest-neighbour connectivity graph of a point cloud (what ``KNNBuilder`` yields for
    ``coord_type="generic"``, /root/reference/src/squidpy/gr/neighbors.py:157-209, before any symmetrisation): row i holds
    ones at its k nearest other points.  Searched on the device (``sqgr_knn_self``); SURVEY.md §8d's graph for config 3 and
    the full-edge-list case of the nhood bench."""
    from . import _lib

    n = xy.shape[0]
    _, idx = _lib.knn_self(ctx if ctx is not None else _lib.default_context(), np.asarray(xy, dtype=np.float64), k)
    g = sparse.csr_matrix((np.ones(n * k, dtype=np.float32), idx.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
    g.sort_indices()
    return g


def hex_adata(rows: int, cols: int, n_cls: int, seed: int = 0, n_genes: int = 0) -> AnnDataLite:
    rng = np.random.default_rng(seed)
    n = rows * cols
    labels = rng.integers(0, n_cls, n)
    X = rng.gamma(2.0, 1.0, size=(n, n_genes)) if n_genes else None
    return AnnDataLite(
        X=X,
        obs=pd.DataFrame({"cluster": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(n_cls)])}),
        obsm={"spatial": hex_grid(rows, cols)},
        obsp={"spatial_connectivities": hex_grid_graph(rows, cols)},
    )
