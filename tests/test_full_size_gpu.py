"""GPU tests at BASELINE.json's full sizes through size-independent properties (plus sampled oracle comparisons)."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import devrng
from oracle import restate as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


def test_c5_nhood_1e6_spots_30_clusters(L, ctx):
    """Config 5 shape: counts bit-exact vs the oracle; every permutation's counts sum to nnz and have the label-degree
    marginals of *some* arrangement of the same multiset; first permutations equal the oracle's generator bit for bit;
    moments equal the sums of the per-permutation counts."""
    rows = cols = 1000
    k = 30
    adj = O.hex_grid_graph(rows, cols)
    labels = np.random.default_rng(0).integers(0, k, rows * cols).astype(np.int32)
    g = L.Graph(ctx, adj, with_data=False)
    count = L.nhood_counts(ctx, g, labels, k)
    np.testing.assert_array_equal(count, O.nhood_counts(adj.indices, adj.indptr, labels, k))
    plan = L.NhoodPlan(ctx, g, labels, k)
    P = 1000
    s1, s2, perms = plan.run(2024, 0, P, None, return_perms=True)
    assert (perms.reshape(P, -1).sum(1) == adj.nnz).all()
    p64 = perms.astype(np.int64)
    np.testing.assert_array_equal(s1, p64.sum(0))
    np.testing.assert_array_equal(s2, (p64 * p64).sum(0).astype(np.uint64))
    for p in (0, 1, 999):
        shuffled = devrng.shuffled_labels(labels, 2024, p)
        assert np.array_equal(np.bincount(shuffled, minlength=k), np.bincount(labels, minlength=k))
        np.testing.assert_array_equal(perms[p], O.nhood_counts(adj.indices, adj.indptr, shuffled, k))
    # split invariance at full size (what multi-GPU sharding relies on)
    a1, a2, _ = plan.run(2024, 0, 300)
    b1, b2, _ = plan.run(2024, 300, P)
    np.testing.assert_array_equal(a1 + b1, s1)
    np.testing.assert_array_equal(a2 + b2, s2)
    # null moments: mean of count[a,b] under shuffling ~ nnz * p_a * p_b
    freq = np.bincount(labels, minlength=k) / len(labels)
    np.testing.assert_allclose(p64.mean(0), adj.nnz * np.outer(freq, freq), rtol=0.01)


def test_c4_cooccurrence_1e6_points(L, ctx):
    """Config 4 shape (1e6 points, 30 clusters, 49 radii): N(N-1) ordered pairs within an infinite radius, symmetry
    under label transposition, monotone cumulative counts, row totals = m_a (N-1); sampled sub-problem == oracle."""
    rows = cols = 1000
    n, k = rows * cols, 30
    rng = np.random.default_rng(4)
    xy = (O.hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))).astype(np.float32)
    labs = rng.integers(0, k, n).astype(np.int32)
    tmin, tmax = O.find_min_max(xy)
    interval = np.linspace(tmin, tmax, num=50, dtype=np.float32)
    thr = np.append(interval[1:] ** 2, np.float32(np.inf)).astype(np.float32)
    got = L.cooccur_counts(ctx, xy[:, 0], xy[:, 1], labs, k, thr)
    assert got[..., -1].sum() == n * (n - 1)
    m = np.bincount(labs, minlength=k).astype(np.int64)
    np.testing.assert_array_equal(got[..., -1].sum(1), m * (n - 1))
    np.testing.assert_array_equal(got, np.transpose(got, (1, 0, 2)))
    assert (np.diff(got, axis=2) >= 0).all()
    sel = np.where(labs < 2)[0][:1200]
    sub = L.cooccur_counts(ctx, xy[sel, 0], xy[sel, 1], labs[sel], 2, thr)
    np.testing.assert_array_equal(sub, O.occur_count(xy[sel, 0], xy[sel, 1], thr, labs[sel], 2))


def test_c3_autocorr_1e5_spots(L, ctx):
    """Config 3 shape per gene block (1e5 spots, k=6 graph, row-normalised): sampled genes/permutations == oracle at
    rtol 1e-6; permutation null has mean -1/(N-1) (Moran) / 1 (Geary)."""
    from sklearn.preprocessing import normalize

    rows, cols, G, P = 250, 400, 192, 64
    n = rows * cols
    g = normalize(O.hex_grid_graph(rows, cols), norm="l1", axis=1)
    rng = np.random.default_rng(1)
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    vals[:8] += np.sin(O.hex_grid(rows, cols)[:, 0] / 500.0)
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    sel = [0, 5, 100, 191]
    for mode, func, null in (("moran", O.morans_i, -1.0 / (n - 1)), ("geary", O.gearys_c, 1.0)):
        np.testing.assert_allclose(plan.scores(mode)[sel], func(g, vals[sel]), rtol=1e-6, atol=1e-12)
        sims = plan.perms(mode, seed=3, perm_begin=0, perm_end=P)
        idx = np.stack([devrng.autocorr_permutation(n, 3, p) for p in (0, 63)])
        np.testing.assert_allclose(sims[[0, 63]][:, sel], O.score_perms(mode, g, vals[sel], idx), rtol=1e-6, atol=1e-12)
        assert abs(sims.mean() - null) < 1e-3
    assert plan.scores("moran")[:8].min() > 0.05  # the spatially structured genes stand out


def test_c4_ripley_l_1e6_points(L, ctx):
    """Config 4 Ripley L: per-cluster pair counts at full size; the largest radius covers every pair; one cluster
    checked against sklearn's KDTree (what the reference calls)."""
    from sklearn.neighbors import KDTree

    rows = cols = 1000
    n, k = rows * cols, 30
    rng = np.random.default_rng(4)
    xy = O.hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
    labs = rng.integers(0, k, n)
    support = np.append(np.linspace(0, 60000, 49), 1e9)
    pts = xy[labs == 7]
    pc = L.pair_counts(ctx, pts, support)
    assert pc[-1] == len(pts) * (len(pts) - 1) and (np.diff(pc) >= 0).all()
    ref = KDTree(pts).two_point_correlation(pts, support[:12], dualtree=True) - len(pts)
    np.testing.assert_array_equal(pc[:12], ref)
