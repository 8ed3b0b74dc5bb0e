"""Spot order and the neighbourhood-enrichment count kernel (no counterpart in the reference: its numba loop does not care).

The count kernel gathers the 16-byte label rows of an edge's two endpoints; when neighbouring spots lie near each other in
``obs`` order (a grid in scan order, cells listed field of view by field of view) the rows of consecutive edges share cache lines.
Spots in NO spatial order cost up to 8x in that kernel (1e6 spots, 30 clusters, ``tools/spot_order_time.py``: 9.0 ms instead of
1.1 ms per 2560 permutations; the headline's 885 k permutations/s become 240 k) — and a bandwidth-reducing renumbering brings all
of it back (882 k).  :func:`spatial_order` computes such an order on the host; :func:`edge_span` is the cheap diagnostic the front
end uses to point the problem out.  Renumbering ``obs`` changes no statistic of the test, but it changes WHICH arrangement a seed
draws (both generators permute positions), so the library never does it behind the caller's back."""

from __future__ import annotations

from typing import Any

import numpy as np


def edge_span(adj: Any, sample: int = 200_000) -> float:
    """Mean ``|row - col| / n`` over (a sample of) the stored edges of a CSR matrix: ~1/3 for spots in random order,
    ``~1 / sqrt(n)`` for a 2-D grid in scan order."""
    n, nnz = adj.shape[0], int(adj.nnz)
    if n < 2 or nnz == 0:
        return 0.0
    indptr, indices = np.asarray(adj.indptr), np.asarray(adj.indices)
    if nnz <= sample:
        e = np.arange(nnz)
    else:
        e = np.random.default_rng(0).integers(0, nnz, sample)
    rows = np.searchsorted(indptr, e, side="right") - 1
    return float(np.abs(indices[e].astype(np.int64) - rows).mean() / n)


def spatial_order(adj: Any = None, coords: Any = None) -> np.ndarray:
    """A renumbering ``order`` (new position -> old index) under which neighbouring spots are close in memory: the Z-order
    (Morton) curve of ``coords`` (n x 2, e.g. ``adata.obsm['spatial']``) when given, else the reverse Cuthill-McKee order of the
    graph ``adj``.  Use as ``adata = adata[order].copy()`` (AnnData) before building the graph, or permute ``obs`` / ``obsp``
    alike — ~0.1 s per million spots."""
    if coords is not None:
        xy = np.asarray(coords, dtype=np.float64)[:, :2]
        lo, hi = xy.min(axis=0), xy.max(axis=0)
        q = ((xy - lo) / np.maximum(hi - lo, np.finfo(np.float64).tiny) * 65535.0).astype(np.uint64)

        def spread(v: np.ndarray) -> np.ndarray:  # 16 bits -> every second bit of 32
            v = (v | (v << 8)) & np.uint64(0x00FF00FF)
            v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
            v = (v | (v << 2)) & np.uint64(0x33333333)
            return (v | (v << 1)) & np.uint64(0x55555555)

        return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)), kind="stable")
    if adj is None:
        raise ValueError("spatial_order needs the graph `adj` or the coordinates `coords`.")
    from scipy import sparse
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    a = sparse.csr_matrix(adj)
    return np.asarray(reverse_cuthill_mckee(a, symmetric_mode=False), dtype=np.int64)
