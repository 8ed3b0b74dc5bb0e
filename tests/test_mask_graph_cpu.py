"""`mask_graph` (host code, gr/_build.py:852-954): the segment-within-polygon predicate against a dense-sampling oracle, and
the reference's own test (tests/graph/test_spatial_neighbors.py:388-456) on its fixture (tests/conftest.py:412-441)."""

from __future__ import annotations

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import squidpy_amd as sq
from squidpy_amd.gr._mask import MultiPolygon, Polygon, _classify, _polygons, segments_within


def _within_by_sampling(a, b, mask, n=4001):
    """The definition, literally: no sampled point of the segment outside, some sampled interior point strictly inside."""
    t = np.linspace(0.0, 1.0, n)[:, None]
    cls = _classify(a[None, :] + t * (b - a)[None, :], _polygons(mask))
    return bool((cls > 0).all() and (cls[1:-1] == 2).any())


def test_point_classification_known_cases():
    sq_ = Polygon([(0, 0), (4, 0), (4, 4), (0, 4)], holes=[[(1, 1), (2, 1), (2, 2), (1, 2)]])
    pts = np.array([(3, 3), (1.5, 1.5), (0, 2), (2, 1.5), (5, 5), (1, 1), (4, 4), (0.5, 0.5)], dtype=float)
    np.testing.assert_array_equal(_classify(pts, _polygons(sq_)), [2, 0, 1, 1, 0, 1, 1, 2])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_segments_within_matches_dense_sampling(seed):
    rng = np.random.default_rng(seed)
    # a concave polygon with a hole, and a second polygon next to it
    m = Polygon([(0, 0), (4, 0), (4, 8), (2, 1.5), (0, 8)])
    holed = Polygon([(5, 0), (9, 0), (9, 6), (5, 6)], holes=[[(6, 2), (8, 2), (8, 4), (6, 4)]])
    mask = MultiPolygon([m, holed])
    a = rng.uniform((-1, -1), (10, 9), (400, 2))
    b = a + rng.normal(0, 1.5, (400, 2))
    got = segments_within(a, b, mask, chunk=97)
    want = np.array([_within_by_sampling(a[k], b[k], mask) for k in range(len(a))])
    np.testing.assert_array_equal(got, want)
    assert got.any() and (~got).any()
    # hand-checked: through the notch of the M, through the hole, along an edge, a chord touching a vertex from inside
    a2 = np.array([(1, 5), (5.5, 3), (0, 0), (1, 1), (1, 1), (2, 1.4)], dtype=float)
    b2 = np.array([(3, 5), (8.5, 3), (4, 0), (3, 1), (2, 1.5), (2, 1.4)], dtype=float)
    np.testing.assert_array_equal(segments_within(a2, b2, mask), [False, False, False, True, True, True])


def _fixture():
    """The reference's `sdata_mask_graph` fixture (tests/conftest.py:412-441) as a table inside a SpatialData-like object."""
    rng = np.random.default_rng(42)
    points = np.concatenate([rng.uniform((3.2, 4.2), (3.8, 5.2), (3, 2)), rng.uniform((0.2, 4.2), (0.8, 5.2), (3, 2)),
                             rng.uniform((1, 0.5), (3, 1.5), (3, 2)), rng.uniform((1, 5), (2, 6), (3, 2))])
    adata = sq.AnnDataLite(X=rng.normal(size=(len(points), 20)), obs=pd.DataFrame(index=[str(i) for i in range(len(points))]),
                           obsm={"spatial": points})

    class SData:  # `tables` mapping and no `obs`: what extract_adata_if_sdata recognises
        def __init__(self, table):
            self.tables = {"table": table}

    return SData(adata), Polygon([(0, 0), (4, 0), (4, 8), (2, 1.5), (0, 8)])


@pytest.mark.parametrize("key_added", ["mask", "mask2"])
def test_mask_graph_ported_from_reference(key_added):
    sdata, polygon = _fixture()
    table = sdata.tables["table"]
    xy = table.obsm["spatial"]
    n = len(xy)
    # a complete graph with distances (the reference builds one with spatial_neighbors; any graph with both slots serves)
    d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    conn = sp.csr_matrix((d > 0).astype(np.float64))
    table.obsp["spatial_connectivities"] = conn
    table.obsp["spatial_distances"] = sp.csr_matrix(d)
    mask_conns_key, mask_dists_key, mask_neighs_key = f"{key_added}_spatial_connectivities", f"{key_added}_spatial_distances", f"{key_added}_spatial_neighbors"
    assert sq.gr.mask_graph(sdata, "table", polygon, negative_mask=False, key_added=key_added) is None
    original = table.obsp["spatial_connectivities"].copy()
    positive = table.obsp[mask_conns_key].copy()
    sq.gr.mask_graph(sdata, "table", polygon, negative_mask=True, key_added=key_added)
    negative = table.obsp[mask_conns_key].copy()
    assert original.toarray().sum() == positive.toarray().sum() + negative.toarray().sum() == n * (n - 1)
    assert mask_conns_key in table.obsp and mask_dists_key in table.obsp and mask_neighs_key in table.uns
    uns = table.uns[mask_neighs_key]
    assert uns["distances_key"] == mask_dists_key and uns["connectivities_key"] == mask_conns_key and uns["params"]["negative_mask"]
    assert uns["unfiltered_graph_key"] == "spatial_connectivities" and uns["params"]["table_key"] == "table"
    # the two point clouds in the arms of the M are not connected through the notch, and no edge leaves the polygon
    pos = positive.toarray()
    assert pos[:3, 3:6].sum() == 0 and pos[:3, :3].sum() == 6 and pos[3:6, 3:6].sum() == 6
    assert (pos == pos.T).all() and 0 < pos.sum() < n * (n - 1)
    np.testing.assert_array_equal(table.obsp[mask_dists_key].toarray() > 0, negative.toarray() > 0)
    # copy=True returns the pair and writes nothing new
    adj, dst = sq.gr.mask_graph(sdata, "table", polygon, key_added="other", copy=True)
    np.testing.assert_array_equal(adj.toarray(), pos)
    assert "other_spatial_connectivities" not in table.obsp and dst.nnz == adj.nnz
    with pytest.raises(ValueError, match="`polygon_mask` should be of type `Polygon` or `MultiPolygon`, got"):
        sq.gr.mask_graph(sdata, "table", (0, 1), negative_mask=True, key_added=key_added)
    with pytest.raises(TypeError, match="table_key"):
        sq.gr.mask_graph(sdata, None, polygon)
