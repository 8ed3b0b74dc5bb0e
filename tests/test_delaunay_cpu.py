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
