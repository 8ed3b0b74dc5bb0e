"""Graph-builder classes: the reference's one extension API on this path (``squidpy.gr.neighbors``,
/root/reference/src/squidpy/gr/neighbors.py:54-419; docs/extensibility.md) and its entry point
``spatial_neighbors_from_builder`` (gr/_build.py:388-452).

``GraphBuilder`` is the protocol custom builders implement — ``build_graph(coords) -> (adj, dst)`` and ``uns_params()``,
optionally ``postprocessors()`` and ``combine(mats, ixs)`` for ``library_key`` — exactly as upstream, with the reusable post-build
steps (``DistanceIntervalPostprocessor``, ``PercentilePostprocessor``, ``TransformPostprocessor``) and the ``Transform`` enum on
``builder.transform``.  The four built-in
builders (``KNNBuilder``, ``RadiusBuilder``, ``DelaunayBuilder``, ``GridBuilder``) take the reference's constructor
arguments and produce the reference's matrices, but their neighbour searches run in ``libsqgr.so`` (device cell list,
``csrc/sqgr_neighbors.hip``) through ``squidpy_amd.gr._build`` — they are what ``spatial_neighbors_knn`` etc. execute."""

from __future__ import annotations

import warnings
from abc import ABC, abstractmethod
from collections.abc import Sequence
from typing import Any

import numpy as np
from scipy import sparse

from collections.abc import Callable
from dataclasses import dataclass
from typing import TypeVar

from .._constants import Transform
from .._utils import assert_positive

__all__ = ["GraphMatrixT", "GraphBuilder", "GraphBuilderCSR", "GraphPostprocessor", "DistanceIntervalPostprocessor", "PercentilePostprocessor",
           "TransformPostprocessor", "KNNBuilder", "RadiusBuilder", "DelaunayBuilder", "GridBuilder"]

# the matrix type a builder produces, and a post-build step on it (gr/neighbors.py:49-51)
GraphMatrixT = TypeVar("GraphMatrixT")
GraphPostprocessor = Callable[[GraphMatrixT, GraphMatrixT], tuple[GraphMatrixT, GraphMatrixT]]


class GraphBuilder(ABC):
    """Base class for spatial graph construction strategies (gr/neighbors.py:54-106)."""

    def __init__(self, transform: Any = None, set_diag: bool = False, percentile: float | None = None, postprocessors: Sequence[Any] = ()) -> None:
        self.transform = Transform.NONE if transform is None else Transform(getattr(transform, "value", transform))
        self.set_diag = set_diag
        self.percentile = percentile
        self._postprocessors = list(postprocessors)

    def build(self, coords: Any) -> tuple[Any, Any]:
        adj, dst = self.build_graph(coords)
        for post in self.postprocessors():
            adj, dst = post(adj, dst)
        return adj, dst

    @abstractmethod
    def build_graph(self, coords: Any) -> tuple[Any, Any]:
        """Construct raw adjacency and distance matrices."""

    def postprocessors(self) -> Sequence[Any]:
        """Post-build processing steps ``(adj, dst) -> (adj, dst)``."""
        return self._postprocessors

    @abstractmethod
    def uns_params(self) -> dict[str, Any]:
        """Parameters stored in ``adata.uns`` after graph construction."""

    def combine(self, mats: Sequence[tuple[Any, Any]], ixs: Sequence[int]) -> tuple[Any, Any]:
        raise NotImplementedError("Using `library_key` with this graph builder is not implemented yet.")


class GraphBuilderCSR(GraphBuilder, ABC):
    """CSR-based strategy: sparse-efficiency warnings silenced, block-diagonal ``library_key`` combination
    (gr/neighbors.py:109-154)."""

    def build(self, coords: np.ndarray) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", sparse.SparseEfficiencyWarning)
            return super().build(coords)

    def combine(self, mats: Sequence[tuple[sparse.csr_matrix, sparse.csr_matrix]], ixs: Sequence[int]) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
        adj = sparse.block_diag([m[0] for m in mats], format="csr")
        dst = sparse.block_diag([m[1] for m in mats], format="csr")
        ixs_arr = np.asarray(ixs)
        if ixs_arr.size and np.any(np.diff(ixs_arr) < 0):  # interleaved libraries: back to observation order
            order = np.argsort(ixs_arr)
            adj, dst = adj[order, :][:, order], dst[order, :][:, order]
        return sparse.csr_matrix(adj), sparse.csr_matrix(dst)


# ---- the reusable post-build steps custom builders compose (gr/neighbors.py:427-477); each works on the pair in place like the
# reference's and returns it
@dataclass(frozen=True)
class DistanceIntervalPostprocessor:
    """Edges whose length lies outside ``interval = (min, max)`` are zeroed in both matrices; the adjacency diagonal is kept."""

    interval: tuple[float, float]

    def __call__(self, adj: sparse.csr_matrix, dst: sparse.csr_matrix) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
        lo, hi = self.interval
        outside = (dst.data < lo) | (dst.data > hi)
        diagonal = adj.diagonal()
        dst.data[outside] = 0.0
        adj.data[outside] = 0.0
        adj.setdiag(diagonal)
        return adj, dst


@dataclass(frozen=True)
class PercentilePostprocessor:
    """Edges longer than the given percentile of the stored distances are zeroed in both matrices."""

    percentile: float

    def __call__(self, adj: sparse.csr_matrix, dst: sparse.csr_matrix) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
        threshold = np.percentile(dst.data, self.percentile)
        too_long = dst > threshold
        adj[too_long] = 0.0
        dst[too_long] = 0.0
        return adj, dst


@dataclass(frozen=True)
class TransformPostprocessor:
    """Explicit zeros are dropped from both matrices, then the adjacency is transformed (spectral: D^-1/2 A D^-1/2; cosine)."""

    transform: Transform

    def __call__(self, adj: sparse.csr_matrix, dst: sparse.csr_matrix) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
        from ._build import _spectral

        adj.eliminate_zeros()
        dst.eliminate_zeros()
        kind = Transform.NONE if self.transform is None else Transform(getattr(self.transform, "value", self.transform))
        if kind == Transform.SPECTRAL:
            return _spectral(adj if sparse.isspmatrix_csr(adj) else sparse.csr_matrix(adj)), dst
        if kind == Transform.COSINE:
            from sklearn.metrics.pairwise import cosine_similarity

            return cosine_similarity(adj, dense_output=False), dst
        return adj, dst


class _DeviceBuilder(GraphBuilderCSR):
    """A built-in builder: the whole recipe (search on the device, pruning, transform) is `_build._build_one(spec)`."""

    _device: int | None = None

    def _spec(self) -> Any:
        raise NotImplementedError

    def build_graph(self, coords: np.ndarray) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
        from .._lib import default_context
        from ._build import _build_one

        spec = self._spec()
        host_only = spec.kind == "delaunay" or (spec.kind == "grid" and spec.delaunay)
        ctx = None if host_only else default_context(self._device)
        return _build_one(ctx, np.ascontiguousarray(np.asarray(coords)), spec)

    def postprocessors(self) -> Sequence[Any]:
        return ()  # interval / percentile pruning and the transform are part of `_build_one`

    def uns_params(self) -> dict[str, Any]:
        return self._spec().uns_params()


class KNNBuilder(_DeviceBuilder):
    """k-nearest-neighbour graph (gr/neighbors.py:157-209)."""

    def __init__(self, n_neighs: int = 6, transform: Any = None, set_diag: bool = False, percentile: float | None = None) -> None:
        assert_positive(n_neighs, name="n_neighs")
        super().__init__(transform=transform, set_diag=set_diag, percentile=percentile)
        self.n_neighs = n_neighs

    def _spec(self) -> Any:
        from ._build import _Spec

        return _Spec("knn", n_neighs=self.n_neighs, transform=self.transform.value, set_diag=self.set_diag, percentile=self.percentile)


class RadiusBuilder(_DeviceBuilder):
    """Fixed-radius graph; a tuple keeps the edges inside ``[min, max]`` (gr/neighbors.py:212-269)."""

    def __init__(self, radius: float | tuple[float, float], transform: Any = None, set_diag: bool = False, percentile: float | None = None) -> None:
        super().__init__(transform=transform, set_diag=set_diag, percentile=percentile)
        self.radius = radius

    def _spec(self) -> Any:
        from ._build import _Spec

        return _Spec("radius", radius=self.radius, transform=self.transform.value, set_diag=self.set_diag, percentile=self.percentile)


class DelaunayBuilder(_DeviceBuilder):
    """Delaunay triangulation graph (Qhull on the host, as in the reference; gr/neighbors.py:272-332)."""

    def __init__(self, radius: float | tuple[float, float] | None = None, transform: Any = None, set_diag: bool = False,
                 percentile: float | None = None) -> None:
        if isinstance(radius, (int, float)):
            radius = (0.0, float(radius))
        super().__init__(transform=transform, set_diag=set_diag, percentile=percentile)
        self.radius = radius

    def _spec(self) -> Any:
        from ._build import _Spec

        return _Spec("delaunay", radius=self.radius, transform=self.transform.value, set_diag=self.set_diag, percentile=self.percentile)


class GridBuilder(_DeviceBuilder):
    """Grid (Visium-like lattice) graph with optional rings (gr/neighbors.py:335-419)."""

    def __init__(self, n_neighs: int = 6, n_rings: int = 1, delaunay: bool = False, transform: Any = None, set_diag: bool = False) -> None:
        assert_positive(n_neighs, name="n_neighs")
        assert_positive(n_rings, name="n_rings")
        super().__init__(transform=transform, set_diag=set_diag, percentile=None)
        self.n_neighs, self.n_rings, self.delaunay = n_neighs, n_rings, delaunay

    def _spec(self) -> Any:
        from ._build import _Spec

        return _Spec("grid", n_neighs=self.n_neighs, n_rings=self.n_rings, transform=self.transform.value, set_diag=self.set_diag, delaunay=self.delaunay)
