"""Pins ``oracle/restate.py`` (the CPU oracle) against the golden vectors generated from the
reference's own kernel source (tests/golden/make_golden.py) and against the known answers the
reference's tests hold for this path.  CPU only."""

from __future__ import annotations

import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ref_shim as R
from oracle import restate as O


def _csr(g, prefix, n=None, dtype=np.float32):
    indptr = g[f"{prefix}_indptr"]
    n = len(indptr) - 1
    data = g[f"{prefix}_data"] if f"{prefix}_data" in g else np.ones(len(g[f"{prefix}_indices"]), dtype)
    return sp.csr_matrix((data, g[f"{prefix}_indices"], indptr), shape=(n, n))


def test_nhood_counts_match_reference_kernel(golden):
    k = int(golden["nhood_k"])
    c = O.nhood_counts(golden["nhood_indices"], golden["nhood_indptr"], golden["nhood_labels"], k)
    assert c.dtype == np.uint32
    np.testing.assert_array_equal(c, golden["nhood_count"])


def test_nhood_perms_and_zscore_match_reference_helper(golden):
    k = int(golden["nhood_k"])
    P = golden["nhood_perms"].shape[0]
    perms = O.nhood_perm_counts_numpy(
        golden["nhood_indices"], golden["nhood_indptr"], golden["nhood_labels"], k, int(golden["nhood_seed"]), P
    )
    np.testing.assert_array_equal(perms, golden["nhood_perms"])
    z = O.nhood_zscore(golden["nhood_count"], perms)
    np.testing.assert_array_equal(z, golden["nhood_zscore"])


def test_nhood_library_shuffle_matches_reference(golden):
    k = int(golden["nhood_k"])
    P = golden["nhood_perms_lib"].shape[0]
    perms = O.nhood_perm_counts_numpy(
        golden["nhood_indices"],
        golden["nhood_indptr"],
        golden["nhood_labels"],
        k,
        int(golden["nhood_seed"]),
        P,
        lib_codes=golden["nhood_lib_codes"],
        n_libs=3,
    )
    np.testing.assert_array_equal(perms, golden["nhood_perms_lib"])


def test_interaction_matrix_known_answers(golden):
    """reference tests/graph/test_nhood.py:153-173 ([[5,1],[2,3]] weighted, [[4,1],[2,2]] unweighted)."""
    args = (golden["intmat_data"], golden["intmat_indices"], golden["intmat_indptr"], golden["intmat_cats"], 2)
    np.testing.assert_array_equal(O.interaction_matrix(*args, weights=True), [[5, 1], [2, 3]])
    np.testing.assert_array_equal(O.interaction_matrix(*args, weights=False), [[4, 1], [2, 2]])
    # the same gather through the nhood kernel (binarised) must give the unweighted KAT
    c = O.nhood_counts(golden["intmat_indices"], golden["intmat_indptr"], golden["intmat_cats"], 2)
    np.testing.assert_array_equal(c, [[4, 1], [2, 2]])


@pytest.mark.parametrize("name", ["lattice", "jitter"])
def test_cooccurrence_matches_reference_kernel(golden, name):
    xy = golden[f"cooc_{name}_xy"].astype(np.float32)
    labs = golden[f"cooc_{name}_labs"]
    interval = golden[f"cooc_{name}_interval"]
    counts = O.occur_count(xy[:, 0], xy[:, 1], interval[1:] ** 2, labs, 4)
    np.testing.assert_array_equal(counts, golden[f"cooc_{name}_counts"])
    occ = O.co_occurrence_probs(counts)
    np.testing.assert_array_equal(occ, golden[f"cooc_{name}_occ"])
    occ2, iv2 = O.co_occurrence(golden[f"cooc_{name}_xy"], labs, interval=12)
    np.testing.assert_array_equal(iv2, interval)
    np.testing.assert_array_equal(occ2, golden[f"cooc_{name}_occ"])


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_autocorr_pvalues_match_reference(golden, mode):
    g = _csr(golden, "autocorr_g")
    n = g.shape[0]
    score, sims = golden[f"unpinned_{mode}_score"], golden[f"unpinned_{mode}_sims"]
    expected = -1.0 / (n - 1) if mode == "moran" else 1.0
    res = O.p_value_calc(score, sims, g, mode, expected, False)
    for key in ("pval_norm", "pval_z_sim", "pval_sim", "var_sim"):
        np.testing.assert_allclose(res[key], golden[f"autocorr_{mode}_{key}"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(res["var_norm"], golden[f"autocorr_{mode}_var_norm"], rtol=1e-13)
    np.testing.assert_allclose(O.g_moments(g), golden["autocorr_moments"], rtol=1e-13)
    # restated score_perms == literal _score_helper driven by the same (unpinned) statistic
    perm_idx = O.autocorr_perm_indices(n, int(golden["autocorr_seed"]), sims.shape[0])
    np.testing.assert_array_equal(perm_idx, golden["autocorr_perm_idx"])
    np.testing.assert_allclose(O.score_perms(mode, g, golden["autocorr_vals"], perm_idx), sims, rtol=1e-12)


def test_var_norm_closed_form(golden):
    """reference tests/graph/test_ppatterns.py:108-137: var_norm equals the Cliff & Ord closed forms."""
    g = _csr(golden, "autocorr_g")
    n = g.shape[0]
    s0, s1, s2 = O.g_moments(g)
    v_moran = (n * n * s1 - n * s2 + 3 * s0 * s0) / ((n - 1) * (n + 1) * s0 * s0) - (1.0 / (n - 1)) ** 2
    v_geary = ((2 * s1 + s2) * (n - 1) - 4 * s0 * s0) / (2 * (n + 1) * s0 * s0)
    score = golden["unpinned_moran_score"]
    np.testing.assert_allclose(O.analytic_pval(score, g, "moran", -1 / (n - 1), False)[1], v_moran, rtol=1e-10)
    np.testing.assert_allclose(O.analytic_pval(score, g, "geary", 1.0, False)[1], v_geary, rtol=1e-10)
    assert not np.isclose(v_moran, v_geary)


def test_moran_geary_dense_textbook():
    """Self-check of the (parity-unpinned) statistic against the dense definitions."""
    rng = np.random.default_rng(3)
    n = 40
    W = (rng.random((n, n)) < 0.15) * rng.random((n, n))
    np.fill_diagonal(W, 0)
    X = rng.normal(size=(4, n))
    X[2] = 3.0  # constant row -> NaN
    g = sp.csr_matrix(W.astype(np.float32))
    Wd = g.toarray().astype(np.float64)
    I, C = O.morans_i(g, X), O.gearys_c(g, X)
    for k in (0, 1, 3):
        x = X[k]
        z = x - x.mean()
        I_ref = n / Wd.sum() * (Wd * np.outer(z, z)).sum() / (z * z).sum()
        C_ref = (n - 1) * (Wd * (x[:, None] - x[None, :]) ** 2).sum() / (2 * Wd.sum() * (z * z).sum())
        np.testing.assert_allclose(I[k], I_ref, rtol=1e-12)
        np.testing.assert_allclose(C[k], C_ref, rtol=1e-12)
    assert np.isnan(I[2]) and np.isnan(C[2])


def test_fdr_bh_known_values():
    p = np.array([0.01, 0.04, 0.03, 0.20, 0.5])
    # hand computation: sorted .01 .03 .04 .2 .5 -> *5/rank -> .05 .075 .0667 .25 .5 -> cummin from right
    exp = np.array([0.05, 0.2 / 3, 0.2 / 3, 0.25, 0.5])
    np.testing.assert_allclose(O.fdr_bh(p), exp, rtol=1e-12)


def test_ripley_helpers_match_reference(golden):
    pts, support = golden["ripley_points"], golden["ripley_support"]
    _, l = O.l_function(pts, support, 400, 2500.0)
    np.testing.assert_array_equal(l, golden["ripley_l"])
    # brute force == KDTree.two_point_correlation - m
    pairs = O.pair_counts_bruteforce(pts, support)
    k_est = (pairs / 400) / (400 / 2500.0)
    np.testing.assert_array_equal(np.sqrt(k_est / np.pi), golden["ripley_l"])
    _, fg = O.f_g_function(golden["ripley_fg_dist"].squeeze(), support)
    np.testing.assert_array_equal(fg, golden["ripley_fg"])
    from scipy.spatial import ConvexHull

    sim = O.ppp(ConvexHull(pts), 1, 50, np.random.default_rng(5))
    np.testing.assert_array_equal(sim, golden["ripley_ppp"])


def test_hex_graph_equals_gridbuilder_logic():
    """hex_grid_graph == kNN(6) + `dist < 1.3 median` (reference gr/neighbors.py:403-416)."""
    from sklearn.neighbors import NearestNeighbors

    rows, cols = 14, 17
    xy = O.hex_grid(rows, cols)
    n = len(xy)
    dist, idx = NearestNeighbors(n_neighbors=7).fit(xy).kneighbors(xy)
    dist, idx = dist[:, 1:], idx[:, 1:]
    keep = dist < 1.3 * np.median(dist)
    rows_i = np.repeat(np.arange(n), 6)[keep.ravel()]
    ref = sp.csr_matrix((np.ones(keep.sum(), np.float32), (rows_i, idx.ravel()[keep.ravel()])), shape=(n, n))
    assert (ref != O.hex_grid_graph(rows, cols)).nnz == 0


@pytest.mark.skipif(not R.available(), reason="reference tree only exists in the build container")
def test_restatement_equals_literal_source_on_fresh_inputs():
    """Direct restate-vs-reference-source comparison on inputs that are not in the golden file."""
    rng = np.random.default_rng(11)
    n, k = 120, 3
    A = sp.random(n, n, density=0.05, format="csr", random_state=5)
    labels = rng.integers(0, k, n).astype(np.uint32)
    fn = R.nhood()["create_function"](k)
    ref = fn(A.indices.astype(np.uint32), A.indptr.astype(np.uint32), labels)
    np.testing.assert_array_equal(O.nhood_counts(A.indices, A.indptr, labels, k), ref)
    pp = R.ppatterns()
    x, y = (rng.random(90) * 30).astype(np.float32), (rng.random(90) * 30).astype(np.float32)
    labs = rng.integers(0, k, 90).astype(np.int32)
    thr = (np.linspace(1, 20, 7, dtype=np.float32)) ** 2
    np.testing.assert_array_equal(
        O.occur_count(x, y, thr, labs, k), pp["_occur_count"](x, y, thr, labs, 90, k, len(thr))
    )


def test_c_port_matches_golden(golden):
    """oracle/c/sqgr_cpu.c (the timed CPU baseline) reproduces the reference kernels' outputs."""
    from oracle import cport

    k = int(golden["nhood_k"])
    c = cport.nenrich(golden["nhood_indices"], golden["nhood_indptr"], golden["nhood_labels"], k)
    np.testing.assert_array_equal(c, golden["nhood_count"])
    np.testing.assert_array_equal(cport.nenrich(golden["nhood_indices"], golden["nhood_indptr"], golden["nhood_labels"], k, parallel=True), c)
    for name in ("lattice", "jitter"):
        xy = golden[f"cooc_{name}_xy"].astype(np.float32)
        cc = cport.occur_count(xy[:, 0], xy[:, 1], golden[f"cooc_{name}_interval"][1:] ** 2, golden[f"cooc_{name}_labs"], 4)
        np.testing.assert_array_equal(cc, golden[f"cooc_{name}_counts"])
    g = _csr(golden, "autocorr_g")
    np.testing.assert_allclose(cport.morans_i(g, golden["autocorr_vals"]), golden["unpinned_moran_score"], rtol=1e-12)
    np.testing.assert_allclose(cport.gearys_c(g, golden["autocorr_vals"]), golden["unpinned_geary_score"], rtol=1e-12)


def _autocorr_kat():
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "autocorr_kat.json")) as fh:
        cases = json.load(fh)
    for rec in cases:
        n = rec["n"]
        g = sp.csr_matrix((np.array([float.fromhex(h) for h in rec["data_hex"]]), np.array(rec["indices"], dtype=np.int32),
                           np.array(rec["indptr"], dtype=np.int32)), shape=(n, n))
        X = np.array([[float.fromhex(h) for h in row] for row in rec["X_hex"]])
        want = {k: np.array([np.nan if h is None else float.fromhex(h) for h in rec[k]]) for k in ("I", "C")}
        yield rec["name"], g, X, want


def test_moran_geary_exact_rational_known_answers():
    """The statistic's arithmetic is third-party (scanpy.metrics, absent): pinned instead to the documented definitions
    evaluated in exact rational arithmetic (tests/golden/make_autocorr_kat.py) on the reference's own 5-node fixture graph
    (tests/conftest.py:177-194) and on its 49-spot Visium fixture.  The restatement must agree to a few ulps."""
    for name, g, X, want in _autocorr_kat():
        np.testing.assert_allclose(O.morans_i(g, X), want["I"], rtol=1e-13, equal_nan=True, err_msg=name)
        np.testing.assert_allclose(O.gearys_c(g, X), want["C"], rtol=1e-13, equal_nan=True, err_msg=name)
