"""GPU parity tests of ripley: pair counts bit-exact vs sklearn's KDTree (what the reference calls), kNN distances
equal to sklearn's NearestNeighbors, whole-function results equal to the oracle's restatement of gr/_ripley.py."""

from __future__ import annotations

import numpy as np
import pandas as pd
import pytest

from oracle import restate as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


def test_pair_counts_golden_l_function(L, ctx, golden):
    pts, support = golden["ripley_points"], golden["ripley_support"]
    pairs = L.pair_counts(ctx, pts, support)
    k_est = (pairs / 400) / (400 / 2500.0)
    np.testing.assert_array_equal(np.sqrt(k_est / np.pi), golden["ripley_l"])  # output of the reference's `_l_function`


@pytest.mark.parametrize("metric", ["euclidean", "manhattan", "chebyshev"])
@pytest.mark.parametrize("m", [2, 255, 256, 257, 3000])
def test_pair_counts_equal_kdtree(L, ctx, metric, m):
    from sklearn.neighbors import KDTree

    rng = np.random.default_rng(m)
    pts = np.round(rng.random((m, 2)) * 40, 1)  # coarse grid of coordinates: many exact distance ties with radii
    support = np.linspace(0, 25, 50)
    ref = KDTree(pts, metric=metric).two_point_correlation(pts, support, dualtree=True) - m
    np.testing.assert_array_equal(L.pair_counts(ctx, pts, support, metric), ref)
    if metric == "euclidean":
        np.testing.assert_array_equal(L.pair_counts(ctx, pts, support), O.pair_counts_bruteforce(pts, support))


@pytest.mark.parametrize("metric", ["euclidean", "manhattan", "chebyshev"])
@pytest.mark.parametrize("k", [1, 2, 3, 5, 9])
def test_knn_equals_sklearn(L, ctx, metric, k):
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(k)
    ref = np.round(rng.random((1500, 2)) * 100, 2)
    qry = np.round(rng.random((700, 2)) * 100, 2)
    exp, _ = NearestNeighbors(metric=metric, n_neighbors=k).fit(ref).kneighbors(qry, n_neighbors=k)
    np.testing.assert_array_equal(L.knn_dist(ctx, qry, ref, k, metric), exp)
    with pytest.raises(ValueError, match="Expected n_neighbors <= n_samples_fit"):
        L.knn_dist(ctx, qry, ref[:1], 2, metric)


def _adata(n=900, k=3, seed=0):
    import squidpy_amd as sq

    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2)) * 500
    lab = rng.integers(0, k, n)
    xy[lab == 0] = xy[lab == 0] * 0.5 + 100  # one clustered population
    obs = pd.DataFrame({"cl": pd.Categorical.from_codes(lab, [f"c{i}" for i in range(k)])})
    return sq.AnnDataLite(obs=obs, obsm={"spatial": xy})


@pytest.mark.parametrize("mode", ["F", "G", "L"])
def test_ripley_equals_reference_restatement(L, mode):
    import squidpy_amd as sq

    adata = _adata()
    res = sq.gr.ripley(adata, "cl", mode=mode, n_simulations=12, n_observations=150, n_steps=20, seed=3, copy=True)
    ref = O.ripley(adata.obsm["spatial"], adata.obs["cl"].values, mode=mode, n_simulations=12, n_observations=150, n_steps=20, seed=3)
    np.testing.assert_array_equal(res["bins"], ref["bins"])
    obs = res[f"{mode}_stat"]
    np.testing.assert_allclose(obs["stats"].to_numpy().reshape(3, 20), ref["obs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(res["sims_stat"]["stats"].to_numpy().reshape(12, 20), ref["sims"], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(res["pvalues"], ref["pvalues"])


def test_ripley_structure_ported_from_reference_tests(L):
    """reference tests/graph/test_ripley.py:13-121."""
    import squidpy_amd as sq

    adata = _adata()
    for mode in ("F", "G", "L"):
        assert sq.gr.ripley(adata, "cl", mode=mode, n_simulations=5, n_observations=80, n_steps=11, seed=1) is None
        res = adata.uns[f"cl_ripley_{mode}"]
        assert set(res) == {f"{mode}_stat", "sims_stat", "bins", "pvalues"}
        obs_df, sims_df = res[f"{mode}_stat"], res["sims_stat"]
        assert res["bins"].shape == (11,) and obs_df.shape == (11 * 3, 3) and sims_df.shape == (11 * 5, 3)
        assert res["pvalues"].shape == (3, 11)
        assert list(obs_df.columns) == ["bins", "cl", "stats"] and list(sims_df.columns) == ["bins", "simulations", "stats"]
        assert obs_df["bins"].iloc[0] == 0 and obs_df["stats"].iloc[0] == 0
        again = sq.gr.ripley(adata, "cl", mode=mode, n_simulations=5, n_observations=80, n_steps=11, seed=1, copy=True)
        pd.testing.assert_frame_equal(again["sims_stat"], sims_df)
        other = sq.gr.ripley(adata, "cl", mode=mode, n_simulations=5, n_observations=80, n_steps=11, seed=2, copy=True)
        assert not np.allclose(other["sims_stat"]["stats"], sims_df["stats"])
        s = sims_df["stats"].to_numpy().reshape(5, 11)
        assert not np.allclose(s[0], s[1])  # simulations differ from each other
    with pytest.raises(ValueError, match="Unsupported metric"):
        sq.gr.ripley(adata, "cl", mode="L", metric="cosine")
    with pytest.raises(ValueError, match="Invalid option `Z` for `RipleyStat`"):
        sq.gr.ripley(adata, "cl", mode="Z")
    res = sq.gr.ripley(adata, "cl", mode="G", metric="manhattan", n_simulations=3, n_observations=50, n_steps=7, seed=0, copy=True, max_dist=100.0)
    assert res["bins"][-1] == 100.0


@pytest.mark.parametrize("metric", ["euclidean", "manhattan", "chebyshev"])
def test_resident_knn_histogram_equals_numpy_histogram(metric):
    """`DevicePoints.knn_hist` == np.histogram(knn distances, bins=edges)[0]: lattice coordinates put many distances
    exactly on bin edges (the last edge is inclusive, everything outside the edges is dropped), one label is excluded."""
    from squidpy_amd import _lib as L

    ctx = L.default_context()
    rng = np.random.default_rng(4)
    q = np.round(rng.random((3000, 2)) * 40)            # integer lattice: exact ties with integer edges
    lab = rng.integers(0, 4, len(q)).astype(np.int32)
    refs = np.round(rng.random((500, 2)) * 40)
    pts = L.DevicePoints(ctx, q, lab)
    dropped = []
    for k, edges, excl in ((1, np.arange(0.0, 6.0), -1), (2, np.linspace(1.0, 9.0, 17), 2), (3, np.array([0.0, 0.5, 0.5, 2.0, 7.0]), 0)):
        got = pts.knn_hist(refs, k, edges, metric, exclude_label=excl)
        keep = np.ones(len(q), bool) if excl < 0 else lab != excl
        dist = L.knn_dist(ctx, q[keep], refs, k, metric)
        want = np.histogram(dist.ravel(), bins=edges)[0]
        np.testing.assert_array_equal(got, want)
        dropped.append(int(dist.size - got.sum()))
    assert max(dropped) > 0  # at least one setting has distances outside its edges
    pts.close()


@pytest.mark.parametrize("metric", ["euclidean", "manhattan", "chebyshev"])
def test_knn_cell_list_outside_queries_duplicates_clusters(L, ctx, metric):
    """Reference sets of 512 points or more go through the cell list (k_knn_cells): two far-apart blobs with duplicated
    points and lattice ties as references, queries spread over a box three times as wide (most of them OUTSIDE the grid,
    like Ripley's F draws them), every register tier of k; equal to sklearn's KD-tree, distance for distance.  The
    resident-query histogram (Ripley's G) goes through the same kernel."""
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(11)
    blob = lambda c, m: np.round(rng.normal(c, 3.0, (m, 2)), 1)
    ref = np.concatenate([blob((0, 0), 900), blob((200, 50), 700), np.repeat(blob((100, 100), 20), 3, axis=0)])
    qry = np.concatenate([rng.uniform(-300, 500, (1500, 2)), ref[:200], np.round(rng.uniform(-10, 10, (300, 2)))])
    for k in (1, 2, 4, 7, 16):
        exp, _ = NearestNeighbors(metric=metric, n_neighbors=k).fit(ref).kneighbors(qry, n_neighbors=k)
        np.testing.assert_array_equal(L.knn_dist(ctx, qry, ref, k, metric), exp)
    # a single row / column of cells (degenerate bounding box) and fewer points than one cell's target
    line = np.stack([np.linspace(0, 100, 600), np.zeros(600)], axis=1)
    exp, _ = NearestNeighbors(metric=metric, n_neighbors=3).fit(line).kneighbors(qry, n_neighbors=3)
    np.testing.assert_array_equal(L.knn_dist(ctx, qry, line, 3, metric), exp)
    lab = rng.integers(0, 3, len(qry)).astype(np.int32)
    pts = L.DevicePoints(ctx, qry, lab)
    edges = np.linspace(0.0, 120.0, 33)
    for k, excl in ((1, -1), (2, 1)):
        keep = np.ones(len(qry), bool) if excl < 0 else lab != excl
        dist = L.knn_dist(ctx, qry[keep], ref, k, metric)
        np.testing.assert_array_equal(pts.knn_hist(ref, k, edges, metric, exclude_label=excl), np.histogram(dist.ravel(), bins=edges)[0])
    pts.close()


@pytest.mark.parametrize("metric", ["canberra"])
def test_ball_tree_metrics_without_parameters(L, ctx, metric):
    """The reference hands `metric` straight to NearestNeighbors (gr/_ripley.py:144,148): canberra — the BallTree metric a bare
    string can name that IS a metric — runs on the device for F / G (sklearn's per-coordinate arithmetic, incl. points on the
    axes and at the origin, where 0/0 terms are skipped), and stays invalid for L like in the reference."""
    from sklearn.neighbors import NearestNeighbors

    import squidpy_amd as sq

    rng = np.random.default_rng(5)
    ref = np.round(rng.normal(0, 30, (900, 2)), 1)
    ref[:5] = [[0, 0], [0, 3], [4, 0], [0, 0], [-2, 0]]
    qry = np.concatenate([np.round(rng.normal(0, 40, (400, 2)), 1), [[0, 0], [0, 5], [3, 0]]])
    for k in (1, 2, 5):
        exp, _ = NearestNeighbors(metric=metric, n_neighbors=k).fit(ref).kneighbors(qry, n_neighbors=k)
        np.testing.assert_array_equal(L.knn_dist(ctx, qry, ref, k, metric), exp)
    adata = _adata()
    res = sq.gr.ripley(adata, "cl", mode="G", metric=metric, n_simulations=4, n_observations=60, n_steps=9, seed=0, copy=True, max_dist=2.0)
    ref_res = O.ripley(adata.obsm["spatial"], adata.obs["cl"].values, mode="G", metric=metric, n_simulations=4, n_observations=60, n_steps=9, seed=0, max_dist=2.0)
    np.testing.assert_allclose(res["G_stat"]["stats"].to_numpy().reshape(3, 9), ref_res["obs"], rtol=1e-12)
    np.testing.assert_array_equal(res["pvalues"], ref_res["pvalues"])
    with pytest.raises(ValueError, match="Unsupported metric"):
        sq.gr.ripley(adata, "cl", mode="L", metric=metric)
    with pytest.raises(NotImplementedError, match="not implemented on the GPU path"):
        sq.gr.ripley(adata, "cl", mode="F", metric="haversine")
    # metrics that need parameters fail in the reference's own sklearn call as well (no argument of `ripley` could carry V / VI)
    for needs_params in ("seuclidean", "mahalanobis"):
        with pytest.raises((TypeError, ValueError)):
            NearestNeighbors(metric=needs_params, n_neighbors=2).fit(ref).kneighbors(qry)
        with pytest.raises(TypeError if needs_params == "seuclidean" else ValueError, match="needs parameters"):  # sklearn's own types
            sq.gr.ripley(adata, "cl", mode="G", metric=needs_params)


@pytest.mark.parametrize("metric", ["euclidean", "chebyshev"])
def test_pair_counts_of_many_sets_in_one_launch(L, ctx, metric):
    """`sqgr_pair_counts_batch` (one launch for all clusters / all simulations of Ripley's L) == the per-set call == sklearn,
    for sets of very different sizes incl. empty, one-point and exactly-one-tile sets."""
    from sklearn.neighbors import KDTree

    rng = np.random.default_rng(21)
    sizes = [0, 1, 2, 255, 256, 257, 1000, 3001, 17]
    sets = [np.round(rng.random((m, 2)) * 30, 1) for m in sizes]
    support = np.linspace(0, 20, 37)
    got = L.pair_counts_batch(ctx, sets, support, metric)
    assert got.shape == (len(sizes), 37)
    for k, pts in enumerate(sets):
        np.testing.assert_array_equal(got[k], L.pair_counts(ctx, pts, support, metric))
        if len(pts) >= 2:
            np.testing.assert_array_equal(got[k], KDTree(pts, metric=metric).two_point_correlation(pts, support, dualtree=True) - len(pts))
    assert L.pair_counts_batch(ctx, [], support).shape == (0, 37)


def test_pair_counts_batch_is_cut_into_launches(L, ctx, monkeypatch):
    """More sets than one launch takes (65 535: `n_simulations` has no upper limit in the reference) and sets of very different
    sizes (each size class gets its own launch grid): same counts, in the caller's order."""
    rng = np.random.default_rng(4)
    sizes = [40, 5000, 300, 41, 1100, 2, 4999, 700, 39, 0, 260]
    sets = [np.round(rng.random((m, 2)) * 30, 1) for m in sizes]
    support = np.linspace(0, 20, 11)
    want = np.stack([L.pair_counts(ctx, pts, support) for pts in sets])
    monkeypatch.setattr(L, "PAIR_BATCH_MAX_SETS", 2)
    np.testing.assert_array_equal(L.pair_counts_batch(ctx, sets, support), want)
