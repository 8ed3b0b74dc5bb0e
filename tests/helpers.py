"""Shared builders for the test-suite (synthetic inputs of SURVEY.md §8d at test sizes)."""

from __future__ import annotations

import numpy as np
import pandas as pd
import scipy.sparse as sp

from oracle import restate as O
from squidpy_amd import AnnDataLite


def knn_graph(xy: np.ndarray, k: int = 6) -> sp.csr_matrix:
    from sklearn.neighbors import NearestNeighbors

    n = len(xy)
    idx = NearestNeighbors(n_neighbors=k + 1).fit(xy).kneighbors(xy, return_distance=False)[:, 1:]
    g = sp.csr_matrix((np.ones(n * k, np.float32), idx.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
    g.sort_indices()
    return g


def hex_adata(rows: int, cols: int, n_cls: int, seed: int = 0, n_genes: int = 0, n_libs: int = 0) -> AnnDataLite:
    rng = np.random.default_rng(seed)
    xy = O.hex_grid(rows, cols)
    n = len(xy)
    g = O.hex_grid_graph(rows, cols)
    labels = rng.integers(0, n_cls, n)
    obs = {"cluster": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(n_cls)])}
    if n_libs:
        obs["library"] = pd.Categorical.from_codes(rng.integers(0, n_libs, n), [f"lib{i}" for i in range(n_libs)])
    X = rng.gamma(2.0, 1.0, size=(n, n_genes)) if n_genes else None
    return AnnDataLite(
        X=X,
        obs=pd.DataFrame(obs),
        obsm={"spatial": xy},
        obsp={"spatial_connectivities": g, "spatial_distances": g * 100.0},
    )


def codes(adata: AnnDataLite, key: str) -> np.ndarray:
    return adata.obs[key].cat.codes.to_numpy().astype(np.int32)
