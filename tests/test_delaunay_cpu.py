"""Delaunay graph builders (gr/neighbors.py:272-331, GridBuilder(delaunay=True) :395-398).  The triangulation is
Qhull's on the host in the reference and here alike, so these run without a GPU; the graphs must equal the oracle's
literal restatement of the builders (setdiag / interval / percentile / transform chain)."""

from __future__ import annotations

import numpy as np
import pandas as pd
import pytest

import squidpy_amd as sq
from oracle import restate as O
from squidpy_amd import AnnDataLite


def _adata(n=400, seed=0, libs=0):
    rng = np.random.default_rng(seed)
    obs = pd.DataFrame(index=[str(i) for i in range(n)])
    if libs:
        obs["lib"] = pd.Categorical(rng.integers(0, libs, n).astype(str))
    return AnnDataLite(obs=obs, obsm={"spatial": rng.random((n, 2)) * 100.0})


def _same(a, b):
    a, b = a.tocsr(), b.tocsr()
    a.sort_indices(); b.sort_indices()
    assert a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)


@pytest.mark.parametrize(
    "kw",
    [{}, {"set_diag": True}, {"radius": 12.0}, {"radius": (3.0, 9.0)}, {"percentile": 90.0}, {"radius": (2.0, 15.0), "percentile": 80.0},
     {"transform": "spectral"}, {"transform": "cosine", "set_diag": True}],
)
def test_delaunay_builder_equals_reference_restatement(kw):
    adata = _adata()
    res = sq.gr.spatial_neighbors_delaunay(adata, copy=True, **kw)
    adj, dst = O.spatial_graph(adata.obsm["spatial"], "delaunay", **kw)
    if kw.get("transform") == "cosine":
        np.testing.assert_allclose(res.connectivities.toarray(), adj.toarray(), rtol=1e-6)
    else:
        _same(res.connectivities, adj)
    _same(res.distances, dst)
    assert res.distances.dtype == np.float64 and (res.distances.diagonal() == 0).all()


@pytest.mark.parametrize("n_rings,set_diag", [(1, False), (1, True), (2, False), (3, True)])
def test_grid_builder_with_delaunay_base(n_rings, set_diag):
    adata = _adata(n=300, seed=2)
    res = sq.gr.spatial_neighbors_grid(adata, delaunay=True, n_rings=n_rings, set_diag=set_diag, copy=True)
    adj, dst = O.spatial_graph(adata.obsm["spatial"], "grid", n_rings=n_rings, set_diag=set_diag, delaunay=True)
    _same(res.connectivities, adj)
    _same(res.distances, dst)


def test_slots_params_and_libraries():
    adata = _adata(n=300, seed=3, libs=3)
    assert sq.gr.spatial_neighbors_delaunay(adata, radius=(1.0, 30.0), library_key="lib", key_added="tri") is None
    assert adata.uns["tri_neighbors"]["params"] == {"coord_type": "generic", "radius": [1.0, 30.0], "transform": None}
    conn = adata.obsp["tri_connectivities"]
    codes = adata.obs["lib"].cat.codes.to_numpy()
    rows, cols = conn.nonzero()
    assert (codes[rows] == codes[cols]).all()  # no edge crosses a library
    for c in range(3):
        m = np.where(codes == c)[0]
        adj, _ = O.spatial_graph(adata.obsm["spatial"][m], "delaunay", radius=(1.0, 30.0))
        _same(conn[m][:, m], adj)


def test_legacy_dispatcher_delaunay_rules():
    adata = _adata(n=200, seed=4)
    with pytest.warns(FutureWarning, match="deprecated"):
        a = sq.gr.spatial_neighbors(adata, coord_type="generic", delaunay=True, radius=5.0, copy=True)  # scalar radius ignored
    adj, dst = O.spatial_graph(adata.obsm["spatial"], "delaunay")
    _same(a.connectivities, adj)
    _same(a.distances, dst)
    with pytest.warns(FutureWarning, match="`n_neighs` is ignored when `delaunay=True`"):
        b = sq.gr.spatial_neighbors(adata, coord_type="generic", delaunay=True, radius=(0.0, 8.0), n_neighs=4, copy=True)
    adj, dst = O.spatial_graph(adata.obsm["spatial"], "delaunay", radius=(0.0, 8.0))
    _same(b.connectivities, adj)
    _same(b.distances, dst)


def test_spatial_neighbors_from_builder_with_a_custom_builder():
    """The reference's extension API (gr/neighbors.py:54-106, gr/_build.py:388-452, docs/extensibility.md): any object with
    `build` / `uns_params` / `combine`.  A custom host-side builder (everything within distance 1.5) and the built-in
    Delaunay builder (host-only: Qhull) through `spatial_neighbors_from_builder`, with and without `library_key`."""
    import pandas as pd
    import scipy.sparse as sp

    import squidpy_amd as sq
    from squidpy_amd.gr.neighbors import DelaunayBuilder, GraphBuilder, GraphBuilderCSR

    class Within(GraphBuilderCSR):
        def __init__(self, r):
            super().__init__()
            self.r = r

        def build_graph(self, coords):
            d = np.sqrt(((coords[:, None, :] - coords[None, :, :]) ** 2).sum(-1))
            adj = sp.csr_matrix(((d <= self.r) & (d > 0)).astype(np.float32))
            return adj, sp.csr_matrix(adj.multiply(d))

        def uns_params(self):
            return {"coord_type": "generic", "radius": self.r, "transform": None}

    rng = np.random.default_rng(0)
    xy = np.stack(np.meshgrid(np.arange(6.0), np.arange(5.0)), -1).reshape(-1, 2)
    obs = pd.DataFrame({"lib": pd.Categorical(rng.integers(0, 2, len(xy)).astype(str))})
    adata = sq.AnnDataLite(obs=obs, obsm={"spatial": xy})
    res = sq.gr.spatial_neighbors_from_builder(adata, Within(1.5), copy=True)
    assert res.connectivities.shape == (30, 30) and res.connectivities[0].nnz == 3  # corner: right, up, diagonal
    sq.gr.spatial_neighbors_from_builder(adata, Within(1.5), key_added="near")
    assert adata.uns["near_neighbors"]["params"]["radius"] == 1.5 and "near_connectivities" in adata.obsp
    # libraries: block-diagonal, no edge between libraries, observation order restored
    lib = sq.gr.spatial_neighbors_from_builder(adata, Within(1.5), library_key="lib", copy=True).connectivities.tocoo()
    codes = obs["lib"].cat.codes.to_numpy()
    assert (codes[lib.row] == codes[lib.col]).all() and lib.nnz > 0
    same = res.connectivities.tocoo()
    keep = codes[same.row] == codes[same.col]
    assert lib.nnz == int(keep.sum())
    # a builder without `combine` cannot do libraries
    class NoCombine(GraphBuilder):
        def build_graph(self, coords):
            return sp.identity(len(coords), format="csr"), sp.identity(len(coords), format="csr")

        def uns_params(self):
            return {}

    with pytest.raises(NotImplementedError, match="library_key"):
        sq.gr.spatial_neighbors_from_builder(adata, NoCombine(), library_key="lib", copy=True)
    # the built-in Delaunay builder == the function that wraps it
    a = sq.gr.spatial_neighbors_from_builder(adata, DelaunayBuilder(radius=3.0), copy=True)
    b = sq.gr.spatial_neighbors_delaunay(adata, radius=3.0, copy=True)
    assert (a.connectivities != b.connectivities).nnz == 0 and (a.distances != b.distances).nnz == 0


def test_reusable_postprocessors_and_the_transform_enum():
    """The extension API of gr/neighbors.py:33-51,442-477 (docs/extensibility.md): custom builders compose the public post-build steps
    and read `self.transform` as the `Transform` enum.  Each step against plain numpy on a small dense matrix."""
    import warnings

    from scipy import sparse

    import squidpy_amd as sq
    from squidpy_amd._constants import Transform
    from squidpy_amd.gr import neighbors as nb

    assert set(nb.__all__) >= {"GraphMatrixT", "GraphPostprocessor", "DistanceIntervalPostprocessor", "PercentilePostprocessor", "TransformPostprocessor"}
    assert Transform(None) is Transform.NONE and Transform("cosine") is Transform.COSINE and Transform.NONE.value is None and Transform.SPECTRAL.s == "spectral"
    with pytest.raises(ValueError, match="Invalid option `foo` for `Transform`"):
        Transform("foo")
    assert nb.KNNBuilder(transform="spectral").transform is Transform.SPECTRAL and nb.KNNBuilder().transform is Transform.NONE
    assert nb.KNNBuilder(transform=Transform.COSINE).uns_params()["transform"] == "cosine"

    rng = np.random.default_rng(0)
    xy = rng.random((30, 2)) * 10
    d = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(-1))
    near = (d < 4.0) & ~np.eye(30, dtype=bool)

    def fresh(diag=False):  # both matrices share one sparsity structure, as the reference's builders produce them (set_diag: an
        stored = near | (np.eye(30, dtype=bool) if diag else False)  # explicit 1 / 0 on the diagonal)
        r, c = np.nonzero(stored)
        adj = sparse.csr_matrix((np.ones(len(r)), (r, c)), shape=(30, 30))
        dst = sparse.csr_matrix((d[r, c], (r, c)), shape=(30, 30))
        return adj, dst

    adj, dst = nb.DistanceIntervalPostprocessor((1.0, 3.0))(*fresh(diag=True))
    keep = near & (d >= 1.0) & (d <= 3.0)
    np.testing.assert_array_equal(adj.toarray(), keep.astype(float) + np.eye(30))  # the diagonal survives
    np.testing.assert_array_equal(dst.toarray(), np.where(keep, d, 0.0))

    adj, dst = fresh()
    thr = np.percentile(dst.data, 60.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", sparse.SparseEfficiencyWarning)
        adj, dst = nb.PercentilePostprocessor(60.0)(adj, dst)
    np.testing.assert_array_equal(adj.toarray(), (near & (d <= thr)).astype(float))
    np.testing.assert_array_equal(dst.toarray(), np.where(near & (d <= thr), d, 0.0))

    adj, dst = fresh()
    adj.data[:3] = 0.0
    a2, d2 = nb.TransformPostprocessor(Transform.SPECTRAL)(adj, dst)
    dense = adj.toarray()
    deg = dense.sum(0)
    with np.errstate(divide="ignore"):
        want = dense * np.sqrt(1.0 / deg)[:, None] * np.sqrt(1.0 / deg)[None, :]
    np.testing.assert_allclose(a2.toarray(), np.nan_to_num(want, posinf=0.0), rtol=1e-6)
    assert a2.dtype == np.float32 and (a2.data != 0).all()  # explicit zeros were dropped first
    from sklearn.metrics.pairwise import cosine_similarity

    a3, _ = nb.TransformPostprocessor("cosine")(*fresh())
    np.testing.assert_allclose(a3.toarray(), cosine_similarity(near.astype(float)), rtol=1e-12)
    a4, _ = nb.TransformPostprocessor(Transform.NONE)(*fresh())
    np.testing.assert_array_equal(a4.toarray(), near.astype(float))

    class Composed(nb.GraphBuilderCSR):  # a custom builder in the documented style
        def __init__(self, radius, **kw):
            super().__init__(postprocessors=[nb.DistanceIntervalPostprocessor((0.0, radius)), nb.TransformPostprocessor(Transform(kw.get("transform")))], **kw)

        def build_graph(self, coords):
            dd = np.sqrt(((coords[:, None] - coords[None]) ** 2).sum(-1))
            m = (dd < 4.0) & ~np.eye(len(coords), dtype=bool)
            return sparse.csr_matrix(m.astype(np.float64)), sparse.csr_matrix(np.where(m, dd, 0.0))

        def uns_params(self):
            return {"transform": self.transform.value}

    adata = sq.AnnDataLite(obs=pd.DataFrame(index=[str(i) for i in range(30)]), obsm={"spatial": xy})
    res = sq.gr.spatial_neighbors_from_builder(adata, Composed(2.5, transform="spectral"), copy=True)
    keep = near & (d <= 2.5)
    deg = keep.sum(0).astype(float)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = keep * np.sqrt(1.0 / deg)[:, None] * np.sqrt(1.0 / deg)[None, :]
    np.testing.assert_allclose(res.connectivities.toarray(), np.nan_to_num(want, posinf=0.0), rtol=1e-6)
