"""Spot order and the neighbourhood-enrichment count kernel (no counterpart in the reference: its numba loop does not care).

The count kernel gathers the 16-byte label rows of an edge's two endpoints; when neighbouring spots lie near each other in
``obs`` order (a grid in scan order, cells listed field of view by field of view) the rows of consecutive edges share cache lines.
Spots in NO spatial order cost up to 8x in that kernel (1e6 spots, 30 clusters, ``tools/spot_order_time.py``: 9.0 ms instead of
1.1 ms per 2560 permutations; the headline's 885 k permutations/s become 240 k) — and a bandwidth-reducing renumbering brings all
of it back (882 k).  ``rng="philox"`` therefore runs its plan on a renumbered TWIN of the graph when :func:`edge_locality` says the
order is not spatial (gr/_nhood.py: ``_internal_order``; the order comes from ``obsm['spatial']`` along the Z-order curve, computed
on the device, or from :func:`spatial_order` of the graph): the device generator permutes the ranks of the CALLER's observations, so
the moments are the ones of the plan on the caller's own graph, bit for bit.  numpy's streams permute POSITIONS of the label vector
— their rows would have to be gathered into the renumbered slab, one random byte per label — and keep the caller's order; a caller
who wants the default stream fast on unordered data renumbers ``obs`` itself (:func:`spatial_order`), knowing that a seed then draws
another arrangement."""

from __future__ import annotations

from typing import Any

import numpy as np


def _sampled_offsets(adj: Any, sample: int) -> np.ndarray:
    """``|row - col|`` of the stored edges of 64 evenly spaced runs of consecutive rows (about ``sample`` edges; all of them on a
    small matrix) — a milliseconds' work on a million rows (random probes into ``indptr`` cost 30 ms)."""
    n, nnz = adj.shape[0], int(adj.nnz)
    indptr, indices = np.asarray(adj.indptr), np.asarray(adj.indices)
    if nnz <= sample:
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
        return np.abs(indices.astype(np.int64) - rows)
    run = max(1, int(sample / max(nnz / n, 1e-9)) // 64)
    out = []
    for r0 in np.linspace(0, n - run, 64).astype(np.int64):
        r1 = int(r0) + run
        e0, e1 = int(indptr[r0]), int(indptr[r1])
        rows = np.repeat(np.arange(r0, r1, dtype=np.int64), np.diff(indptr[r0 : r1 + 1]))
        out.append(np.abs(indices[e0:e1].astype(np.int64) - rows))
    return np.concatenate(out) if out else np.zeros(1, dtype=np.int64)


def edge_span(adj: Any, sample: int = 200_000) -> float:
    """Mean ``|row - col| / n`` over (a sample of) the stored edges of a CSR matrix: ~1/3 for spots in random order,
    ``~1 / sqrt(n)`` for a 2-D grid in scan order."""
    n, nnz = adj.shape[0], int(adj.nnz)
    if n < 2 or nnz == 0:
        return 0.0
    return float(_sampled_offsets(adj, sample).mean() / n)


def edge_locality(adj: Any, sample: int = 200_000) -> tuple[float, float]:
    """``(near, span)`` over (a sample of) the stored edges: the fraction whose endpoints lie within 8 positions of each other —
    they share a 128-byte line of 16-byte label rows; a third of a grid's edges in scan order, next to none when the cells of a
    tile come in random order — and :func:`edge_span`."""
    n, nnz = adj.shape[0], int(adj.nnz)
    if n < 2 or nnz == 0:
        return 1.0, 0.0
    d = _sampled_offsets(adj, sample)
    return float((d <= 8).mean()), float(d.mean() / n)


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
