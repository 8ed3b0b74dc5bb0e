"""BASELINE.json's five configurations, WHOLE, through the front end (`squidpy_amd.gr.*`) at their stated sizes and
`n_perms`.  Where the oracle cannot be run at full size in seconds it is run on an exact sub-problem
(a sample of permutations / genes, clusters small enough for brute force) and size-independent identities cover the rest."""

from __future__ import annotations

import numpy as np
import pandas as pd
import pytest
from contextlib import nullcontext as _nullcontext

from oracle import devrng
from oracle import restate as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq():
    import squidpy_amd

    return squidpy_amd


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


def _adata(sq, rows, cols, labels, k, X=None, xy=None, graph=True):
    obs = pd.DataFrame({"cluster": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(k)])})
    return sq.AnnDataLite(X=X, obs=obs, obsm={"spatial": O.hex_grid(rows, cols) if xy is None else xy},
                          obsp={"spatial_connectivities": O.hex_grid_graph(rows, cols)} if graph else {})


def test_config1_nhood_5000_spots_10_clusters_1000_perms(sq):
    """Config 1 (the reference's own CPU-runnable case) in full, both generators: every one of the 1000 permutations
    against the oracle — the DEFAULT call (numpy's streams on the device): Squidpy's z-scores for the seed, bit for bit;
    rng="philox": the oracle's restatement of the device generator."""
    rows, cols, k, P = 50, 100, 10, 1000
    labels = np.random.default_rng(0).integers(0, k, rows * cols).astype(np.int32)
    adata = _adata(sq, rows, cols, labels, k)
    adj = adata.obsp["spatial_connectivities"]
    count = O.nhood_counts(adj.indices, adj.indptr, labels, k)
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=42, copy=True)
    np.testing.assert_array_equal(res.counts, count)
    np.testing.assert_array_equal(res.zscore, O.nhood_zscore(count, O.nhood_perm_counts_numpy(adj.indices, adj.indptr, labels, k, 42, P)))
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=42, copy=True, rng="philox")
    np.testing.assert_array_equal(res.counts, count)
    np.testing.assert_allclose(res.zscore, O.nhood_zscore(count, O.nhood_perm_counts_philox(adj.indices, adj.indptr, labels, k, 42, 0, P)), rtol=1e-9)
    # and through the AnnData slot (gr/_nhood.py:236-242)
    sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=42, rng="philox")
    np.testing.assert_array_equal(adata.uns["cluster_nhood_enrichment"]["zscore"], res.zscore)
    np.testing.assert_array_equal(adata.uns["cluster_nhood_enrichment"]["count"], count)


def test_config2_nhood_1e5_spots_20_clusters_10000_perms(sq, L):
    """Config 2 in full: n_perms = 10 000 through the front end; the z-score equals the one formed from all 10 000
    per-permutation counts (C ABI), a 256-permutation sample of which equals the oracle bit for bit."""
    rows, cols, k, P, seed = 250, 400, 20, 10_000, 7
    labels = np.random.default_rng(2).integers(0, k, rows * cols).astype(np.int32)
    adata = _adata(sq, rows, cols, labels, k)
    adj = adata.obsp["spatial_connectivities"]
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=seed, copy=True, rng="philox")
    np.testing.assert_array_equal(res.counts, O.nhood_counts(adj.indices, adj.indptr, labels, k))
    ctx = L.default_context()
    g = L.Graph(ctx, adj, with_data=False)
    plan = L.NhoodPlan(ctx, g, labels, k)
    _, _, perms = plan.run(seed, 0, P, return_perms=True)
    np.testing.assert_allclose(res.zscore, O.nhood_zscore(res.counts, perms), rtol=1e-9)
    assert (perms.reshape(P, -1).sum(1) == adj.nnz).all()
    sample = np.r_[0:200, 4990:5018, 9972:10000]   # 256 permutations incl. both ends and a launch-group boundary
    for p in sample[::8]:
        np.testing.assert_array_equal(perms[p], O.nhood_counts(adj.indices, adj.indptr, devrng.shuffled_labels(labels, seed, int(p)), k))
    ref = np.stack([O.nhood_counts(adj.indices, adj.indptr, s, k)
                    for s in np.sort(labels)[devrng.label_permutations(len(labels), seed, sample)]])
    np.testing.assert_array_equal(perms[sample], ref)
    plan.close()
    g.close()


@pytest.fixture(scope="module")
def config3(sq):
    """1e5 spots x 20 000 genes of float64 expression (16 GB), kNN-6 directed graph (KNNBuilder logic): built once."""
    from tests.helpers import knn_graph

    rows, cols, G = 250, 400, 20_000
    n = rows * cols
    xy = O.hex_grid(rows, cols)
    g = knn_graph(xy + np.random.default_rng(5).normal(0, 1.0, xy.shape), 6)
    rng = np.random.default_rng(1)
    X = np.empty((n, G), dtype=np.float64)
    for r0 in range(0, n, 5000):  # uniform(0, 4): cheap to draw at 2e9 values; spatial structure on 10 % of the genes
        blk = X[r0 : r0 + 5000]
        rng.random(out=blk)
        blk *= 4.0
        blk[:, ::10] += np.sin(xy[r0 : r0 + 5000, 0] / 500.0)[:, None]
    const = [137, 19_999]
    X[:, const] = 3.25
    names = np.array([f"g{i}" for i in range(G)])
    adata = sq.AnnDataLite(X=X, obs=pd.DataFrame(index=[f"s{i}" for i in range(n)]), var=pd.DataFrame(index=names),
                           obsm={"spatial": xy}, obsp={"spatial_connectivities": g})
    return adata, g, names, const


@pytest.mark.parametrize("mode,rng_mode", [("moran", "philox"), ("geary", "numpy")])
def test_config3_autocorr_1e5_spots_20000_genes_1000_perms(sq, L, config3, mode, rng_mode):
    """Config 3 in full (16 GB of float64 expression, kNN-6 directed graph, row-normalised, 1000 permutations, both
    statistics, both permutation sources): sampled genes — structured, plain, constant (NaN) — carry every column of the
    oracle's frame at rtol 1e-6 over ALL 1000 permutations; the rest through identities (null mean, sorted, FDR bounds)."""
    from sklearn.preprocessing import normalize

    adata, g, names, const = config3
    n, G = adata.shape
    P, seed = 1000, 11
    with pytest.warns(UserWarning):
        df = sq.gr.spatial_autocorr(adata, mode=mode, n_perms=P, seed=seed, copy=True, rng=rng_mode)
    stat = "I" if mode == "moran" else "C"
    assert df.shape == (G, 9) and df[stat].isna().sum() == 2 and set(df.index[df[stat].isna()]) == set(names[const])
    ok = df[stat].dropna()
    assert ok.is_monotonic_decreasing if mode == "moran" else ok.is_monotonic_increasing
    sel = [0, 10, 137, 5000, 7777, 12345, 19990, 19999] + ([9990, 15000, 19998, 3] if mode == "moran" else [])
    if rng_mode == "philox":  # the device generator's permutations (parity hook), a sample of which == the oracle's restatement
        idx = L.autocorr_perm_indices(L.default_context(), n, seed, 0, P)
        for p in (0, 1, 500, 999):
            np.testing.assert_array_equal(idx[p], devrng.autocorr_permutation(n, seed, p))
    else:
        idx = O.autocorr_perm_indices(n, seed, P)
    gn = normalize(g.astype(np.float64), norm="l1", axis=1)
    vals = np.ascontiguousarray(adata.X[:, sel].T)
    score = (O.morans_i if mode == "moran" else O.gearys_c)(gn, vals)
    sims = O.score_perms(mode, gn, vals, idx)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = O.p_value_calc(score, sims, gn, mode, -1.0 / (n - 1) if mode == "moran" else 1.0, False)  # gr/_ppatterns.py:218-222
    got = df.loc[names[sel]]
    np.testing.assert_allclose(got[stat].to_numpy(), score, rtol=1e-6, atol=1e-12, equal_nan=True)
    for c, v in ref.items():
        np.testing.assert_allclose(got[c].to_numpy(), v, rtol=1e-6, atol=1e-12, equal_nan=True, err_msg=c)
    # the unstructured genes sit at the null expectation, the structured ones far from it
    plain = df.loc[names[[i for i in range(0, G, 97) if i % 10 and i not in const]], stat].to_numpy()
    assert abs(plain.mean() - (-1.0 / (n - 1) if mode == "moran" else 1.0)) < 2e-3
    struct = df.loc[names[10:2000:10], stat].to_numpy()
    assert (struct > 0.02).all() if mode == "moran" else (struct < 0.98).all()
    # Benjamini-Hochberg over the whole column, NaN p-values of the constant genes included (they propagate, as in statsmodels)
    for c in ("pval_norm", "pval_sim", "pval_z_sim"):
        np.testing.assert_allclose(df[f"{c}_fdr_bh"].to_numpy(), O.fdr_bh(df[c].to_numpy()), rtol=1e-12, equal_nan=True)


def test_config3_sparse_float32_expression_at_full_shape(sq, L, config3):
    """VERDICT r2 task 2: the input every real Visium / Xenium object has — scipy CSR float32, here 1e5 spots x 20 000 genes
    at 10 % density (2e8 stored counts) — is uploaded as it is and densified on the device.  (i) Bit-identical to the dense
    float64 path: the frame of 4096 of the genes equals, in every bit, the frame of their dense float64 form; (ii) end to end
    (upload included, 1000 permutations) the sparse call takes at most 1.5x the time of the dense call on the same shape."""
    import time

    import scipy.sparse as sp

    adata, g, names, const = config3
    n, G = adata.shape
    rng = np.random.default_rng(7)
    parts = []
    for r0 in range(0, n, 2000):  # unique columns per row: a canonical matrix, as scanpy's readers produce
        mask = rng.random((min(2000, n - r0), G)) < 0.1
        blk = sp.csr_matrix(mask, dtype=np.float32)
        blk.data = rng.integers(1, 30, blk.nnz).astype(np.float32)
        parts.append(blk)
    Xs = sp.vstack(parts, format="csr")
    del parts
    assert Xs.dtype == np.float32 and abs(Xs.nnz / (n * G) - 0.1) < 0.005
    sparse_ad = sq.AnnDataLite(X=Xs, obs=adata.obs, var=adata.var, obsp=adata.obsp)
    P, seed = 1000, 11

    def timed(a, **kw):
        best, out = np.inf, None
        for _ in range(2):
            t0 = time.perf_counter()
            with pytest.warns(UserWarning) if kw.get("expect_nan") else _nullcontext():
                out = sq.gr.spatial_autocorr(a, mode="moran", n_perms=P, seed=seed, copy=True, **{k: v for k, v in kw.items() if k != "expect_nan"})
            best = min(best, time.perf_counter() - t0)
        return best, out

    t_dense, _ = timed(adata, expect_nan=True)
    t_sparse, df_sparse = timed(sparse_ad)
    print(f"config 3 end to end: dense float64 {t_dense:.2f} s, CSR float32 (10 % density) {t_sparse:.2f} s")
    assert df_sparse.shape == (G, 9) and np.isfinite(df_sparse["I"]).all()
    assert t_sparse <= 1.5 * t_dense, (t_sparse, t_dense)
    sub = list(names[8000:12096])
    dense_sub = sq.AnnDataLite(X=Xs[:, 8000:12096].toarray().astype(np.float64), obs=adata.obs, var=adata.var.iloc[8000:12096], obsp=adata.obsp)
    a = sq.gr.spatial_autocorr(dense_sub, mode="moran", n_perms=200, seed=seed, copy=True)
    b = sq.gr.spatial_autocorr(sparse_ad, genes=sub, mode="moran", n_perms=200, seed=seed, copy=True)
    pd.testing.assert_frame_equal(a, b, check_exact=True)
    # the statistic of the full call agrees with the subset call (other columns depend on the number of permutations / FDR set)
    np.testing.assert_array_equal(df_sparse.loc[b.index, "I"].to_numpy(), b["I"].to_numpy())


def _occur_count_blocked(x, y, thr2, labels, k, block=10_000):
    """The C restatement of `_occur_count` (gr/_ppatterns.py:283-310; oracle/c/sqgr_cpu.c keeps the reference's per-point int32
    scratch of K*K*L counters: 8.8 GB for 5e4 points) evaluated block-wise by inclusion-exclusion over pairs of point blocks:
    counts(all) = sum_p counts(B_p) + sum_{p<q} [counts(B_p u B_q) - counts(B_p) - counts(B_q)].  Integer counts: exact."""
    from oracle import cport

    n = len(x)
    blocks = [np.arange(b0, min(n, b0 + block)) for b0 in range(0, n, block)]
    own = [cport.occur_count(x[b], y[b], thr2, labels[b], k, parallel=True) for b in blocks]
    total = sum(own)
    for p in range(len(blocks)):
        for q in range(p + 1, len(blocks)):
            u = np.concatenate([blocks[p], blocks[q]])
            total = total + cport.occur_count(x[u], y[u], thr2, labels[u], k, parallel=True) - own[p] - own[q]
    return total


def test_config4_cooccurrence_every_cell_exact_on_a_50000_point_subcloud(sq, L):
    """All 30 x 30 x 49 cells of config 4's count array, not only the small clusters': a 50 000-point sub-cloud of the same point
    set (hex grid + N(0, 5) jitter, 30 uniform labels, the front end's own 49 thresholds) through `sq.gr.co_occurrence` against the
    C restatement of the reference's loop — bit-exact integers, the reference's ratio arithmetic on top."""
    rows = cols = 1000
    n, k, m = rows * cols, 30, 50_000
    rng = np.random.default_rng(4)
    xy_all = O.hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
    pick = np.sort(rng.choice(n, m, replace=False))
    xy = xy_all[pick]
    labels = rng.integers(0, k, m).astype(np.int32)
    adata = _adata(sq, 1, m, labels, k, xy=xy, graph=False)
    occ, interval = sq.gr.co_occurrence(adata, "cluster", interval=50, copy=True)
    sp32 = xy.astype(np.float32)
    np.testing.assert_array_equal(interval, np.linspace(*O.find_min_max(sp32), num=50, dtype=np.float32))
    thr2 = interval[1:] ** 2
    want = _occur_count_blocked(sp32[:, 0], sp32[:, 1], thr2, labels, k)
    got = L.cooccur_counts(L.default_context(), sp32[:, 0], sp32[:, 1], labels, k, thr2)
    np.testing.assert_array_equal(got, want)
    assert int(want[..., -1].sum()) > 0.2 * m * (m - 1)  # the last radius (half the diagonal) reaches a good part of all ordered pairs
    np.testing.assert_allclose(occ, O.co_occurrence_probs(want), rtol=1e-12)


def test_config4_cooccurrence_and_ripley_1e6_points_30_clusters(sq, L):
    """Config 4 through both front ends at 1e6 points (hex grid + N(0, 5) jitter), 30 clusters, interval = 50 / n_steps = 50.
    Two of the 30 clusters are small (1500 points), so every count that involves only them has an exact brute-force oracle
    at the FULL problem size; the co-occurrence ratios are the reference's arithmetic applied to the device counts."""
    rows = cols = 1000
    n, k = rows * cols, 30
    rng = np.random.default_rng(4)
    xy = O.hex_grid(rows, cols) + rng.normal(0, 5, (n, 2))
    labels = rng.integers(0, k - 2, n).astype(np.int32)
    small = rng.choice(n, 3000, replace=False)
    labels[small[:1500]], labels[small[1500:]] = k - 2, k - 1
    adata = _adata(sq, rows, cols, labels, k, xy=xy, graph=False)
    occ, interval = sq.gr.co_occurrence(adata, "cluster", interval=50, copy=True)
    assert occ.shape == (k, k, 49) and interval.shape == (50,) and interval.dtype == np.float32 and np.isfinite(occ).all()
    sp32 = xy.astype(np.float32)
    np.testing.assert_array_equal(interval, np.linspace(*O.find_min_max(sp32), num=50, dtype=np.float32))
    thr2 = interval[1:] ** 2
    counts = L.cooccur_counts(L.default_context(), sp32[:, 0], sp32[:, 1], labels, k, thr2)
    np.testing.assert_allclose(occ, O.co_occurrence_probs(counts), rtol=1e-12)
    sel = np.where(labels >= k - 2)[0]
    sub = O.occur_count(sp32[sel, 0], sp32[sel, 1], thr2, labels[sel] - (k - 2), 2)
    np.testing.assert_array_equal(counts[k - 2 :, k - 2 :], sub)           # exact at full size for the small clusters
    np.testing.assert_array_equal(counts, np.transpose(counts, (1, 0, 2)))
    assert (np.diff(counts, axis=2) >= 0).all() and counts[..., -1].sum() <= n * (n - 1)
    # ---- Ripley L, 50 radii, every cluster; the small clusters and one big one against sklearn's KDTree (the reference's call)
    res = sq.gr.ripley(adata, "cluster", mode="L", n_simulations=20, n_observations=500, n_steps=50, seed=3, copy=True)
    stats = res["L_stat"]
    assert len(stats) == k * 50 and np.isfinite(stats["stats"]).all()
    support = res["bins"]
    from scipy.spatial import ConvexHull

    area = ConvexHull(xy).volume
    for c, upto in ((k - 2, 50), (k - 1, 50), (5, 10)):  # the big cluster: the first 10 radii keep sklearn's dual tree quick
        pts = xy[labels == c]
        _, want = O.l_function(pts, support[:upto], n, area)                     # gr/_ripley.py:212-227 (sklearn KDTree)
        got = stats.loc[stats["cluster"] == f"c{c}", "stats"].to_numpy()
        np.testing.assert_allclose(got[:upto], want, rtol=1e-12)
    assert res["pvalues"].shape == (k, 50)


def test_config5_nhood_1e6_spots_30_clusters_100000_perms(sq, L):
    """Config 5 in full: 100 000 permutations through the front end (single GPU here: the whole range in one rank); z-scores ==
    the ones formed from all 100 000 per-permutation counts; sampled permutations == the oracle; split invariance."""
    rows = cols = 1000
    k, P, seed = 30, 100_000, 2024
    labels = np.random.default_rng(0).integers(0, k, rows * cols).astype(np.int32)
    adata = _adata(sq, rows, cols, labels, k)
    adj = adata.obsp["spatial_connectivities"]
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=P, seed=seed, copy=True, rng="philox")
    np.testing.assert_array_equal(res.counts, O.nhood_counts(adj.indices, adj.indptr, labels, k))
    ctx = L.default_context()
    g = L.Graph(ctx, adj, with_data=False)
    plan = L.NhoodPlan(ctx, g, labels, k)
    s1, s2, perms = plan.run(seed, 0, P, return_perms=True)
    np.testing.assert_allclose(res.zscore, O.nhood_zscore(res.counts, perms), rtol=1e-9)
    assert (perms.reshape(P, -1).sum(1) == adj.nnz).all()
    base = np.sort(labels)
    for p in (0, 17, 65_535, 99_999):
        np.testing.assert_array_equal(perms[p], O.nhood_counts(adj.indices, adj.indptr, base[devrng.label_permutations(len(labels), seed, np.array([p]))[0]], k))
    parts = [plan.run(seed, lo, hi) for lo, hi in ((0, 12_500), (12_500, 50_001), (50_001, P))]  # what 8-GPU sharding relies on
    np.testing.assert_array_equal(sum(p[0] for p in parts), s1)
    np.testing.assert_array_equal(sum(p[1] for p in parts), s2)
    plan.close()
    g.close()
