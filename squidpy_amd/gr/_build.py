"""Spatial graph construction with the neighbour searches on the GPU.

Drop-ins for ``squidpy.gr.spatial_neighbors_knn`` / ``_radius`` / ``_grid`` and the legacy dispatcher
``spatial_neighbors`` (reference: /root/reference/src/squidpy/gr/_build.py:132-328, 484-552, 553-622, 701-786, 789-849;
builders gr/neighbors.py:157-269, 335-419; post-processing gr/neighbors.py:441-476; transforms :479-560).

The k-nearest-neighbour and fixed-radius searches — the part that dominates at 1e6 spots — run in ``libsqgr.so`` on a
device cell list (``csrc/sqgr_neighbors.hip``); the O(nnz) CSR assembly, ring expansion, percentile / interval pruning
and the optional spectral / cosine transforms stay on the host with scipy, written to give the reference's matrices.
Delaunay graphs take their triangulation from ``scipy.spatial.Delaunay`` (Qhull) on the host exactly as the reference does
(gr/neighbors.py:319-331, 395-398) — there is no device kernel behind them; edge lengths, pruning and transforms share
the code of the other builders.
"""

from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import Any, NamedTuple

import numpy as np
from scipy import sparse

from .._lib import Context, default_context, knn_self, radius_self
from .._utils import _assert_categorical_obs, _assert_spatial_basis, _save_data, assert_positive, extract_adata_if_sdata

__all__ = [
    "spatial_neighbors",
    "spatial_neighbors_knn",
    "spatial_neighbors_radius",
    "spatial_neighbors_grid",
    "spatial_neighbors_delaunay",
    "spatial_neighbors_from_builder",
    "SpatialNeighborsResult",
]

_TRANSFORMS = ("spectral", "cosine", None)


class SpatialNeighborsResult(NamedTuple):
    """Result of the spatial_neighbors functions (gr/_build.py:56-60)."""

    connectivities: sparse.csr_matrix
    distances: sparse.csr_matrix


@dataclass
class _Spec:
    """What to build.  ``kind``: "knn" | "radius" | "grid" | "delaunay"."""

    kind: str
    n_neighs: int = 6
    radius: Any = None
    n_rings: int = 1
    transform: str | None = None
    set_diag: bool = False
    percentile: float | None = None
    delaunay: bool = False  # grid kind: base connectivity from the triangulation instead of the kNN candidates

    def uns_params(self) -> dict[str, Any]:
        if self.kind == "grid":
            return {"coord_type": "grid", "n_neighbors": self.n_neighs, "n_rings": self.n_rings, "delaunay": self.delaunay,
                    "transform": self.transform}
        if self.kind in ("radius", "delaunay"):
            rad = list(self.radius) if isinstance(self.radius, tuple) else self.radius
            return {"coord_type": "generic", "radius": rad, "transform": self.transform}
        return {"coord_type": "generic", "n_neighbors": self.n_neighs, "transform": self.transform}


def _check_transform(transform: Any) -> str | None:
    value = getattr(transform, "value", transform)
    if value not in _TRANSFORMS:
        raise ValueError(f"Invalid option `{value}` for `Transform`. Valid options are: `{list(_TRANSFORMS)}`.")
    return value


def _with_diagonal(adj: sparse.csr_matrix, value: float | None) -> sparse.csr_matrix:
    """``adj.setdiag(value)`` semantics without scipy's efficiency warning; ``None`` keeps the stored diagonal."""
    if value is None:
        return adj
    n = adj.shape[0]
    out = adj - sparse.diags(adj.diagonal(), format="csr", dtype=adj.dtype)
    if value:
        out = out + sparse.identity(n, format="csr", dtype=adj.dtype) * adj.dtype.type(value)
    out = sparse.csr_matrix(out)
    out.eliminate_zeros()
    return out


def _knn_edges(ctx: Context, coords: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    dist, idx = knn_self(ctx, coords, k)  # == NearestNeighbors(n_neighbors=k).fit(coords).kneighbors()
    n = coords.shape[0]
    return np.repeat(np.arange(n), k), idx.reshape(-1).astype(np.int64), dist.reshape(-1)


def _delaunay_edges(coords: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(row, col) of the vertex adjacency of Qhull's Delaunay triangulation — host code, as in the reference."""
    from scipy.spatial import Delaunay

    indptr, indices = Delaunay(coords).vertex_neighbor_vertices
    return np.repeat(np.arange(coords.shape[0]), np.diff(indptr)), np.asarray(indices, dtype=np.int64)


def _build_one(ctx: Context | None, coords: np.ndarray, spec: _Spec) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
    """One library: the builder's edges, then its post-processing chain (gr/neighbors.py:74-78), done on flat edge
    arrays (row, col, length, alive) and assembled into CSR once at the end."""
    coords = np.asarray(coords, dtype=np.float64)
    uses_device = spec.kind in ("knn", "radius") or (spec.kind == "grid" and not spec.delaunay)
    if uses_device and (coords.ndim != 2 or coords.shape[1] != 2):
        raise NotImplementedError(f"The GPU neighbour search handles 2-D coordinates, found shape `{coords.shape}`.")
    n = coords.shape[0]
    if spec.kind == "grid":
        adj, dst = _grid_graph(ctx, coords, spec)
    else:
        if spec.kind == "knn":
            rows, cols, length = _knn_edges(ctx, coords, spec.n_neighs)
        elif spec.kind == "delaunay":
            rows, cols = _delaunay_edges(coords)
            length = np.linalg.norm(coords[rows] - coords[cols], axis=1)
        else:
            r = spec.radius if isinstance(spec.radius, (int, float)) else max(spec.radius)
            indptr, cols, length = radius_self(ctx, coords, float(r))
            rows, cols = np.repeat(np.arange(n), np.diff(indptr)), cols.astype(np.int64)
        length = length.copy()
        alive = np.ones(len(rows), dtype=bool)
        interval = None
        if spec.kind == "radius" and isinstance(spec.radius, tuple):
            interval = spec.radius
        elif spec.kind == "delaunay" and spec.radius is not None:  # a scalar r is shorthand for (0, r) (gr/neighbors.py:296-300)
            interval = spec.radius if isinstance(spec.radius, tuple) else (0.0, float(spec.radius))
        if interval is not None:  # interval pruning (gr/neighbors.py:425-438)
            lo, hi = sorted(interval)
            out = (length < lo) | (length > hi)
            length[out] = 0.0
            alive &= ~out
        if spec.percentile is not None:  # gr/neighbors.py:451-459: the reference's `dst.data` holds, besides the edges,
            # the n explicit zeros left by `dst.setdiag(0.0)` and the zeros of already pruned edges
            threshold = np.percentile(np.concatenate([length, np.zeros(n)]), spec.percentile)
            far = length > threshold
            length[far] = 0.0
            alive &= ~far
        adj = sparse.csr_matrix((np.ones(int(alive.sum()), dtype=np.float32), (rows[alive], cols[alive])), shape=(n, n))
        if spec.set_diag:
            adj = _with_diagonal(adj, 1.0)
        has_len = alive & (length != 0.0)  # `eliminate_zeros`: zero-length edges (coincident points) leave `dst`
        dst = sparse.csr_matrix((length[has_len], (rows[has_len], cols[has_len])), shape=(n, n))
    adj, dst = sparse.csr_matrix(adj), sparse.csr_matrix(dst)
    adj.eliminate_zeros()
    dst.eliminate_zeros()
    adj.sort_indices()
    dst.sort_indices()
    if spec.transform == "spectral":
        adj = _spectral(adj)
    elif spec.transform == "cosine":
        from sklearn.metrics.pairwise import cosine_similarity

        adj = cosine_similarity(adj, dense_output=False)
    return adj, dst


def _grid_base(ctx: Context | None, coords: np.ndarray, n_neighs: int, diag: float | None, delaunay: bool = False) -> sparse.csr_matrix:
    """kNN candidates pruned at 1.3 x the median candidate distance (gr/neighbors.py:389-419)."""
    n = coords.shape[0]
    if delaunay:
        rows, cols = _delaunay_edges(coords)
        keep = np.ones(len(rows), dtype=bool)
    else:
        rows, cols, dists = _knn_edges(ctx, coords, n_neighs)
        keep = dists < np.median(dists) * 1.3
    adj = sparse.csr_matrix((np.ones(int(keep.sum()), dtype=np.float32), (rows[keep], cols[keep])), shape=(n, n))
    return _with_diagonal(adj, diag)


def _grid_graph(ctx: Context | None, coords: np.ndarray, spec: _Spec) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
    """Ring distances by repeated sparse products (gr/neighbors.py:366-387)."""
    if spec.n_rings > 1:
        base = _grid_base(ctx, coords, spec.n_neighs, 1.0, spec.delaunay)
        reached, walk = base, base
        for ring in range(2, spec.n_rings + 1):
            walk = sparse.csr_matrix(walk @ base)
            fresh = walk - walk.multiply(reached.astype(bool))  # newly reached at this ring
            fresh = sparse.csr_matrix(fresh)
            fresh.eliminate_zeros()
            fresh.data[:] = float(ring)
            walk = fresh
            reached = sparse.csr_matrix(reached + fresh)
        ringed = _with_diagonal(reached, float(spec.set_diag))
        ringed.eliminate_zeros()
        dst = ringed.copy()
        adj = ringed.copy()
        adj.data[:] = 1.0
    else:
        adj = _grid_base(ctx, coords, spec.n_neighs, 1.0 if spec.set_diag else None, spec.delaunay)
        dst = adj.copy()
    dst = _with_diagonal(dst, 0.0)
    return adj, dst


def _spectral(adj: sparse.csr_matrix) -> sparse.csr_matrix:
    """D^-1/2 A D^-1/2 with D = column sums, float32 (gr/neighbors.py:514-548)."""
    if not adj.nnz:
        return adj
    with np.errstate(divide="ignore"):
        deg = np.squeeze(np.asarray(np.sqrt(1.0 / adj.sum(axis=0))))
    rows = np.repeat(np.arange(adj.shape[0]), np.diff(adj.indptr))
    data = (deg[rows] * deg[adj.indices] * adj.data).astype(np.float32)
    return sparse.csr_matrix((data, adj.indices, adj.indptr), shape=adj.shape)


def _run(adata: Any, spec: _Spec, *, spatial_key: str, library_key: str | None, key_added: str, copy: bool,
         device: int | None) -> SpatialNeighborsResult | None:
    """gr/_build.py:789-849: per-library graphs, block-diagonal combination, slot writes."""
    host_only = spec.kind == "delaunay" or (spec.kind == "grid" and spec.delaunay)  # Qhull on the host, nothing to launch
    ctx = None if host_only else default_context(device)
    coords = np.asarray(adata.obsm[spatial_key])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", sparse.SparseEfficiencyWarning)
        if library_key is not None:
            _assert_categorical_obs(adata, key=library_key)
            codes = adata.obs[library_key].cat.codes.to_numpy()
            mats, order = [], []
            for code in range(len(adata.obs[library_key].cat.categories)):
                members = np.where(codes == code)[0]
                mats.append(_build_one(ctx, np.ascontiguousarray(coords[members]), spec))
                order.extend(members.tolist())
            adj = sparse.block_diag([m[0] for m in mats], format="csr")
            dst = sparse.block_diag([m[1] for m in mats], format="csr")
            order_arr = np.asarray(order)
            if order_arr.size and np.any(np.diff(order_arr) < 0):  # interleaved libraries: back to observation order
                back = np.argsort(order_arr)
                adj, dst = adj[back, :][:, back], dst[back, :][:, back]
            adj, dst = sparse.csr_matrix(adj), sparse.csr_matrix(dst)
        else:
            adj, dst = _build_one(ctx, coords, spec)
    if copy:
        return SpatialNeighborsResult(connectivities=adj, distances=dst)
    conn_key, dist_key = f"{key_added}_connectivities", f"{key_added}_distances"
    _save_data(adata, attr="obsp", key=conn_key, data=adj)
    _save_data(adata, attr="obsp", key=dist_key, data=dst)
    _save_data(adata, attr="uns", key=f"{key_added}_neighbors",
               data={"connectivities_key": conn_key, "distances_key": dist_key, "params": spec.uns_params()})
    return None


def spatial_neighbors_from_builder(
    data: Any,
    builder: Any,
    *,
    spatial_key: str = "spatial",
    elements_to_coordinate_systems: dict[str, str] | None = None,
    table_key: str | None = None,
    library_key: str | None = None,
    key_added: str = "spatial",
    copy: bool = False,
    n_jobs: int = 1,
) -> SpatialNeighborsResult | None:
    """Create a graph from spatial coordinates using an explicit builder instance (drop-in for
    ``squidpy.gr.spatial_neighbors_from_builder``, gr/_build.py:388-452 + `_run_spatial_neighbors` :789-849).

    ``builder`` is any object with the :class:`squidpy_amd.gr.neighbors.GraphBuilder` interface — ``build(coords)``
    returning ``(adj, dst)``, ``uns_params()`` and, for ``library_key``, ``combine(mats, ixs)``; the built-in builders of
    :mod:`squidpy_amd.gr.neighbors` run their neighbour searches on the GPU.  ``n_jobs`` is accepted and ignored."""
    adata = _resolve_input(data, spatial_key, elements_to_coordinate_systems, table_key)
    coords = np.asarray(adata.obsm[spatial_key])
    if library_key is not None:
        _assert_categorical_obs(adata, key=library_key)
        codes = adata.obs[library_key].cat.codes.to_numpy()
        mats, ixs = [], []
        for code in range(len(adata.obs[library_key].cat.categories)):
            members = np.where(codes == code)[0]
            mats.append(builder.build(np.ascontiguousarray(coords[members])))
            ixs.extend(members.tolist())
        adj, dst = builder.combine(mats, ixs)
    else:
        adj, dst = builder.build(coords)
    if copy:
        return SpatialNeighborsResult(connectivities=adj, distances=dst)
    conn_key, dist_key = f"{key_added}_connectivities", f"{key_added}_distances"
    _save_data(adata, attr="obsp", key=conn_key, data=adj)
    _save_data(adata, attr="obsp", key=dist_key, data=dst)
    _save_data(adata, attr="uns", key=f"{key_added}_neighbors",
               data={"connectivities_key": conn_key, "distances_key": dist_key, "params": builder.uns_params()})
    return None


def _resolve_input(data: Any, spatial_key: str, elements_to_coordinate_systems: Any, table_key: str | None) -> Any:
    if elements_to_coordinate_systems is not None:
        raise NotImplementedError("`elements_to_coordinate_systems` (SpatialData element matching) is not supported here.")
    adata = extract_adata_if_sdata(data, table_key=table_key)
    _assert_spatial_basis(adata, spatial_key)
    return adata


def spatial_neighbors_knn(
    data: Any,
    *,
    spatial_key: str = "spatial",
    elements_to_coordinate_systems: dict[str, str] | None = None,
    table_key: str | None = None,
    library_key: str | None = None,
    n_neighs: int = 6,
    percentile: float | None = None,
    transform: str | None = None,
    set_diag: bool = False,
    key_added: str = "spatial",
    copy: bool = False,
    n_jobs: int = 1,
    device: int | None = None,
) -> SpatialNeighborsResult | None:
    """Create a k-nearest-neighbor graph from spatial coordinates (drop-in for ``squidpy.gr.spatial_neighbors_knn``,
    gr/_build.py:484-552); the neighbour search runs on the GPU, ties are broken by the smaller observation index."""
    assert_positive(n_neighs, name="n_neighs")
    spec = _Spec("knn", n_neighs=n_neighs, transform=_check_transform(transform), set_diag=set_diag, percentile=percentile)
    adata = _resolve_input(data, spatial_key, elements_to_coordinate_systems, table_key)
    return _run(adata, spec, spatial_key=spatial_key, library_key=library_key, key_added=key_added, copy=copy, device=device)


def spatial_neighbors_radius(
    data: Any,
    *,
    radius: float | tuple[float, float],
    spatial_key: str = "spatial",
    elements_to_coordinate_systems: dict[str, str] | None = None,
    table_key: str | None = None,
    library_key: str | None = None,
    percentile: float | None = None,
    transform: str | None = None,
    set_diag: bool = False,
    key_added: str = "spatial",
    copy: bool = False,
    n_jobs: int = 1,
    device: int | None = None,
) -> SpatialNeighborsResult | None:
    """Create a radius-based graph from spatial coordinates (drop-in for ``squidpy.gr.spatial_neighbors_radius``,
    gr/_build.py:553-622); a tuple ``radius`` builds with the maximum and keeps edges inside the interval."""
    spec = _Spec("radius", radius=radius, transform=_check_transform(transform), set_diag=set_diag, percentile=percentile)
    adata = _resolve_input(data, spatial_key, elements_to_coordinate_systems, table_key)
    return _run(adata, spec, spatial_key=spatial_key, library_key=library_key, key_added=key_added, copy=copy, device=device)


def spatial_neighbors_delaunay(
    data: Any,
    *,
    spatial_key: str = "spatial",
    elements_to_coordinate_systems: dict[str, str] | None = None,
    table_key: str | None = None,
    library_key: str | None = None,
    radius: float | tuple[float, float] | None = None,
    percentile: float | None = None,
    transform: str | None = None,
    set_diag: bool = False,
    key_added: str = "spatial",
    copy: bool = False,
    n_jobs: int = 1,
    device: int | None = None,
) -> SpatialNeighborsResult | None:
    """Create a Delaunay triangulation graph from spatial coordinates (drop-in for ``squidpy.gr.spatial_neighbors_delaunay``,
    gr/_build.py:625-697).  The triangulation is Qhull's (``scipy.spatial.Delaunay`` on the host, as in the reference);
    ``radius`` only prunes the finished edges: ``(min, max)`` keeps lengths inside the interval, a scalar ``r`` means
    ``(0, r)``."""
    spec = _Spec("delaunay", radius=radius, transform=_check_transform(transform), set_diag=set_diag, percentile=percentile)
    adata = _resolve_input(data, spatial_key, elements_to_coordinate_systems, table_key)
    return _run(adata, spec, spatial_key=spatial_key, library_key=library_key, key_added=key_added, copy=copy, device=device)


def spatial_neighbors_grid(
    data: Any,
    *,
    spatial_key: str = "spatial",
    elements_to_coordinate_systems: dict[str, str] | None = None,
    table_key: str | None = None,
    library_key: str | None = None,
    n_neighs: int = 6,
    n_rings: int = 1,
    delaunay: bool = False,
    transform: str | None = None,
    set_diag: bool = False,
    key_added: str = "spatial",
    copy: bool = False,
    n_jobs: int = 1,
    device: int | None = None,
) -> SpatialNeighborsResult | None:
    """Create a grid-based graph from spatial coordinates (drop-in for ``squidpy.gr.spatial_neighbors_grid``,
    gr/_build.py:701-786): kNN candidates pruned at 1.3 x the median distance, ``n_rings`` expansion."""
    assert_positive(n_neighs, name="n_neighs")
    assert_positive(n_rings, name="n_rings")
    spec = _Spec("grid", n_neighs=n_neighs, n_rings=n_rings, transform=_check_transform(transform), set_diag=set_diag,
                 delaunay=bool(delaunay))
    adata = _resolve_input(data, spatial_key, elements_to_coordinate_systems, table_key)
    return _run(adata, spec, spatial_key=spatial_key, library_key=library_key, key_added=key_added, copy=copy, device=device)


def spatial_neighbors(
    adata: Any,
    spatial_key: str = "spatial",
    elements_to_coordinate_systems: dict[str, str] | None = None,
    table_key: str | None = None,
    library_key: str | None = None,
    coord_type: str | None = None,
    n_neighs: int | None = None,
    radius: float | tuple[float, float] | None = None,
    delaunay: bool | None = None,
    n_rings: int | None = None,
    percentile: float | None = None,
    transform: str | None = None,
    set_diag: bool = False,
    key_added: str = "spatial",
    copy: bool = False,
    n_jobs: int = 1,
    *,
    device: int | None = None,
) -> SpatialNeighborsResult | None:
    """Legacy dispatcher (drop-in for the deprecated ``squidpy.gr.spatial_neighbors``, gr/_build.py:132-328): grid mode
    when ``coord_type`` resolves to ``'grid'`` (``None`` + ``adata.uns['spatial']`` present), else radius mode when
    ``radius`` is given, else k-nearest-neighbour mode.  Emits the reference's ``FutureWarning``."""
    warnings.warn(
        "Calling `spatial_neighbors` is deprecated and will be removed in squidpy v1.9.0. Use `spatial_neighbors_knn`, "
        "`spatial_neighbors_radius`, `spatial_neighbors_delaunay`, `spatial_neighbors_grid`, or "
        "`spatial_neighbors_from_builder` instead.",
        FutureWarning,
        stacklevel=2,
    )
    data = _resolve_input(adata, spatial_key, elements_to_coordinate_systems, table_key)
    k = 6 if n_neighs is None else n_neighs
    rings = 1 if n_rings is None else n_rings
    assert_positive(rings, name="n_rings")
    assert_positive(k, name="n_neighs")
    tr = _check_transform(transform)
    if coord_type is None:
        mode = "grid" if "spatial" in data.uns else "generic"  # (the reference logs that `radius` only applies to generic)
    elif coord_type in ("grid", "generic"):
        mode = coord_type
    else:
        raise ValueError(f"Invalid option `{coord_type}` for `CoordType`. Valid options are: `['grid', 'generic']`.")
    if mode == "grid":
        if percentile is not None:
            raise ValueError("`percentile` is not supported for grid coordinates. It only applies to generic (non-grid) graphs.")
        spec = _Spec("grid", n_neighs=k, n_rings=rings, transform=tr, set_diag=bool(set_diag), delaunay=bool(delaunay))
    elif delaunay:
        if n_neighs is not None:
            warnings.warn("Parameter `n_neighs` is ignored when `delaunay=True` use `spatial_neighbors_delaunay` instead.", FutureWarning, stacklevel=2)
        # legacy contract (gr/_build.py:111-117): a scalar `radius` is ignored here, only a tuple prunes
        spec = _Spec("delaunay", radius=radius if isinstance(radius, tuple) else None, transform=tr, set_diag=bool(set_diag),
                     percentile=percentile)
    elif radius is not None:
        if n_neighs is not None:
            warnings.warn("Parameter `n_neighs` is ignored when `radius` is set use `spatial_neighbors_radius` instead.", FutureWarning, stacklevel=2)
        spec = _Spec("radius", radius=radius, transform=tr, set_diag=bool(set_diag), percentile=percentile)
    else:
        spec = _Spec("knn", n_neighs=k, transform=tr, set_diag=bool(set_diag), percentile=percentile)
    return _run(data, spec, spatial_key=spatial_key, library_key=library_key, key_added=key_added, copy=copy, device=device)
