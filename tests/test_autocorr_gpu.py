"""GPU parity tests of spatial_autocorr (Moran's I / Geary's C).  Floating point: rtol 1e-6 (BASELINE north_star),
observed differences are ~1e-13.  NOTE: the statistic itself is "parity unpinned" (scanpy is not in the reference
tree, see oracle/restate.py); everything around it is pinned to the reference's literal source."""

from __future__ import annotations

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from oracle import devrng
from oracle import restate as O
from tests.helpers import knn_graph

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12


@pytest.fixture(autouse=True, params=["gather", "lds", "lds-small-chunks", "lds-split", "lds-split-small-chunks"])
def perm_kernel(request, monkeypatch):
    """Every test of this module runs through the permutation kernels of csrc/sqgr_autocorr.hip: the gather-dot (k_perm_dot), the
    LDS-bucketed dot (k_perm_dot_lds) and the same kernel on 8 virtual permutations per permutation (`lds-split`: what calls with
    fewer than 512 permutations take), the LDS variants also with 64-spot chunks so that small inputs are cut into many (a, b)
    buckets.  The library reads the variables at every call."""
    monkeypatch.setenv("SQGR_AUTOCORR_KERNEL", {"gather": "gather", "lds": "lds", "lds-small-chunks": "lds"}.get(request.param, "lds-split"))
    if request.param.endswith("small-chunks"):
        monkeypatch.setenv("SQGR_AUTOCORR_LDS_CHUNK", "64")
    return request.param


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


def _golden_g(golden):
    n = len(golden["autocorr_g_indptr"]) - 1
    return sp.csr_matrix((golden["autocorr_g_data"], golden["autocorr_g_indices"], golden["autocorr_g_indptr"]), shape=(n, n))


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_scores_and_injected_permutations_match_golden(L, ctx, golden, mode):
    g = _golden_g(golden)
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, golden["autocorr_vals"])
    np.testing.assert_allclose(plan.scores(mode), golden[f"unpinned_{mode}_score"], rtol=RTOL, atol=ATOL)
    # the reference's literal `_score_helper` (g[idx, :] per permutation) vs ONE SpMV + gather-dots on the GPU
    sims = plan.perms(mode, perm_idx=golden["autocorr_perm_idx"])
    np.testing.assert_allclose(sims, golden[f"unpinned_{mode}_sims"], rtol=RTOL, atol=ATOL)


def test_device_permutations_match_oracle_generator(L, ctx):
    for n in (10, 300, 4097):
        got = L.autocorr_perm_indices(ctx, n, seed=77, perm_begin=5, perm_end=9)
        for k, p in enumerate(range(5, 9)):
            np.testing.assert_array_equal(got[k], devrng.autocorr_permutation(n, 77, p))
            assert np.array_equal(np.sort(got[k]), np.arange(n))


@pytest.mark.parametrize("mode", ["moran", "geary"])
@pytest.mark.parametrize("n,G", [(200, 1), (777, 65), (3000, 130)])
def test_vs_oracle_various_shapes(L, ctx, mode, n, G):
    rng = np.random.default_rng(n + G)
    xy = rng.random((n, 2))
    g = knn_graph(xy, 6)
    g.data = rng.random(g.nnz).astype(np.float32) + 0.1  # general (non-normalised, asymmetric) weights
    g = g.tolil()
    g[5, :] = 0  # an isolated row
    g = sp.csr_matrix(g)
    g.eliminate_zeros()
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    vals[0] += 3 * np.sin(xy[:, 0] * 6)
    if G > 2:
        vals[2] = 1.25  # constant feature -> NaN
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    func = O.morans_i if mode == "moran" else O.gearys_c
    np.testing.assert_allclose(plan.scores(mode), func(g, vals), rtol=RTOL, atol=ATOL)
    perm_idx = O.autocorr_perm_indices(n, 3, 21)
    np.testing.assert_allclose(plan.perms(mode, perm_idx=perm_idx), O.score_perms(mode, g, vals, perm_idx), rtol=RTOL, atol=ATOL)
    # device RNG path == oracle evaluated on the device generator's permutations; split invariance
    dev = plan.perms(mode, seed=9, perm_begin=0, perm_end=21)
    idx = np.stack([devrng.autocorr_permutation(n, 9, p) for p in range(21)])
    np.testing.assert_allclose(dev, O.score_perms(mode, g, vals, idx), rtol=RTOL, atol=ATOL)
    two = np.concatenate([plan.perms(mode, seed=9, perm_begin=0, perm_end=8), plan.perms(mode, seed=9, perm_begin=8, perm_end=21)])
    np.testing.assert_array_equal(two, dev)


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_lds_kernel_full_workgroups(L, ctx, mode, perm_kernel):
    """More than 1024 permutations (two permutation blocks, the second with one live wave), three full-size chunks, an odd
    number of features: the LDS-bucketed kernel's staged chunk loads and its list padding against the oracle."""
    if perm_kernel != "lds":
        pytest.skip("default chunking of the LDS kernel only")
    rng = np.random.default_rng(5)
    n, G, P = 12001, 67, 1030
    xy = rng.random((n, 2))
    g = knn_graph(xy, 6)
    g.data = rng.random(g.nnz).astype(np.float32) + 0.1
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    vals[1] += 2 * np.sin(xy[:, 1] * 5)
    vals[4] = -2.0
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    dev = plan.perms(mode, seed=3, perm_begin=2, perm_end=2 + P)
    assert dev.shape == (P, G) and np.isnan(dev[:, 4]).all()
    sel_p = [0, 1, 63, 64, 511, 1023, 1024, 1029]
    idx = np.stack([devrng.autocorr_permutation(n, 3, 2 + p) for p in sel_p])
    np.testing.assert_allclose(dev[sel_p], O.score_perms(mode, g, vals, idx), rtol=RTOL, atol=ATOL)
    # the gather kernel on the same permutations: two independent summation orders of the same sums
    import os

    os.environ["SQGR_AUTOCORR_KERNEL"] = "gather"
    try:
        ref = plan.perms(mode, seed=3, perm_begin=2, perm_end=2 + P)
    finally:
        os.environ["SQGR_AUTOCORR_KERNEL"] = "lds"
    np.testing.assert_allclose(dev, ref, rtol=1e-9, atol=1e-13)
    # ... and the split variant (8 virtual permutations per permutation): a third order of the same sums
    os.environ["SQGR_AUTOCORR_KERNEL"] = "lds-split"
    try:
        ref8 = plan.perms(mode, seed=3, perm_begin=2, perm_end=2 + P)
    finally:
        os.environ["SQGR_AUTOCORR_KERNEL"] = "lds"
    np.testing.assert_allclose(dev, ref8, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_list_schedule_structured_permutations_and_ranges(L, ctx, mode, perm_kernel):
    """The bucket lists of the LDS kernel are re-ordered so that the 16 permutations of a `ds_read_b128` lane group read different LDS
    banks (k_bucket_order_steps / _joint): any order of a list is a valid one, so scores may move by rounding only.  Inputs the
    schedule has to survive: the identity and a stride permutation (every pair of a list in 16 cells), a permutation whose bucket
    (0, 0) holds 320 pairs of ONE cell (i = j = 0 mod 16: the 8-bit cell counters saturate, the segment is left as built), lists of very
    different lengths in one group, and a range of the device generator cut at odd places (a permutation's order depends on the
    15 permutations it shares the lane group with: the library schedules whole aligned groups and drops the extra ones)."""
    if perm_kernel != "lds":
        pytest.skip("default chunking of the LDS kernel only")
    import os

    rng = np.random.default_rng(12)
    n, G = 10216, 9                        # two chunks of 5108 spots
    m = 5108
    xy = rng.random((n, 2))
    g = knn_graph(xy, 6)
    g.data = rng.random(g.nnz).astype(np.float32) + 0.1
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    vals[2] += 2 * np.sin(xy[:, 0] * 5)
    ident = np.arange(n, dtype=np.int32)
    stride = ((np.arange(n, dtype=np.int64) * 4099) % n).astype(np.int32)   # gcd(4099, n) = 1; m = 4 mod 16: classes shift by chunk
    one_cell = ident.copy()
    a0 = np.arange(0, m, 16)                                                 # 320 spots of chunk 0, i = 0 mod 16 -> themselves, reversed
    one_cell[a0] = a0[::-1]
    b0 = np.setdiff1d(np.arange(m), a0)
    b1 = np.setdiff1d(np.arange(m, 2 * m), np.arange(m, 2 * m, 16))[: b0.size]
    one_cell[b0], one_cell[b1] = b1, b0                                      # the rest of chunk 0 <-> chunk 1
    assert np.array_equal(np.sort(one_cell), ident) and b1.size == b0.size
    near = ident.copy()                                                      # almost everything stays inside its chunk: a long and a short list
    near[:16], near[m:m + 16] = np.arange(m, m + 16), np.arange(16)
    perm_idx = np.stack([ident, stride, one_cell, near] + [rng.permutation(n).astype(np.int32) for _ in range(13)])
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    got = plan.perms(mode, perm_idx=perm_idx)
    np.testing.assert_allclose(got, O.score_perms(mode, g, vals, perm_idx), rtol=RTOL, atol=ATOL)
    # the other schedules (the rotation of Z classes; every list on its own, round 3) and the lists as built
    for env, val in (("SQGR_AUTOCORR_ORDER", "rotation"), ("SQGR_AUTOCORR_ORDER", "single"), ("SQGR_AUTOCORR_ORDER_LISTS", "0")):
        os.environ[env] = val
        try:
            np.testing.assert_allclose(plan.perms(mode, perm_idx=perm_idx), got, rtol=1e-10, atol=1e-13)
        finally:
            del os.environ[env]
    # the device generator: [70, 150) alone, in pieces cut inside a lane group, and as part of [0, 200)
    whole = plan.perms(mode, seed=21, perm_begin=0, perm_end=200)
    part = plan.perms(mode, seed=21, perm_begin=70, perm_end=150)
    np.testing.assert_array_equal(part, whole[70:150])
    pieces = np.concatenate([plan.perms(mode, seed=21, perm_begin=b, perm_end=e) for b, e in ((70, 71), (71, 129), (129, 150))])
    np.testing.assert_array_equal(pieces, part)
    idx = np.stack([devrng.autocorr_permutation(n, 21, p) for p in (70, 149)])
    np.testing.assert_allclose(part[[0, -1]], O.score_perms(mode, g, vals, idx), rtol=RTOL, atol=ATOL)
    plan.close()
    graph.close()


def test_range_cuts_with_the_default_kernel_choice(L, ctx, monkeypatch, perm_kernel):
    """the summation kernel is chosen from the length of the CALL's range (gather < 40 <= lds-split < 512 <= lds), so
    bit-identity across range cuts holds for pieces on one side of the thresholds — with the library's own choice, not a forced
    kernel: [0, 1100) == [0, 560) + [560, 1100) (both `lds`), [0, 300) == [0, 130) + [130, 300) (both `lds-split`); a piece
    on the other side agrees to rounding only (documented in include/sqgr.h)."""
    if perm_kernel != "gather":
        pytest.skip("runs once, with no kernel forced")
    monkeypatch.delenv("SQGR_AUTOCORR_KERNEL", raising=False)
    rng = np.random.default_rng(12)
    n, G = 6000, 256
    g = knn_graph(rng.random((n, 2)), 6)
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    whole = plan.perms("moran", seed=5, perm_begin=0, perm_end=1100)
    cut = np.concatenate([plan.perms("moran", seed=5, perm_begin=b, perm_end=e) for b, e in ((0, 560), (560, 1100))])
    np.testing.assert_array_equal(cut, whole)
    small = plan.perms("moran", seed=5, perm_begin=0, perm_end=300)
    cut = np.concatenate([plan.perms("moran", seed=5, perm_begin=b, perm_end=e) for b, e in ((0, 130), (130, 300))])
    np.testing.assert_array_equal(cut, small)
    np.testing.assert_allclose(small, whole[:300], rtol=1e-11, atol=1e-14)   # another kernel: another summation order, the same scores
    plan.close()
    graph.close()


def test_bucket_lists_with_row_sum_classes_belong_to_their_plan(L, ctx, perm_kernel):
    """The lists of the class-table kernel carry the row-sum class of every pair, and the context keeps the lists of the last call:
    two plans of the same size, seed and permutation count on graphs whose spots fall into DIFFERENT classes must not share them."""
    if perm_kernel != "lds":
        pytest.skip("default chunking of the LDS kernel only")
    from sklearn.preprocessing import normalize

    rng = np.random.default_rng(3)
    g1 = normalize(O.hex_grid_graph(70, 70), norm="l1", axis=1)
    g2 = g1.tolil()
    g2[[5, 900, 2000], :] = 0
    g2 = g2.tocsr()
    g2.eliminate_zeros()
    vals = rng.gamma(2.0, 1.0, size=(12, g1.shape[0]))
    out = []
    for g in (g1, g2):
        graph = L.Graph(ctx, g)
        plan = L.AutocorrPlan(ctx, graph, vals)
        out.append(plan.perms("geary", seed=5, perm_begin=0, perm_end=64))
        idx = np.stack([devrng.autocorr_permutation(g.shape[0], 5, p) for p in (0, 63)])
        np.testing.assert_allclose(out[-1][[0, 63]], O.score_perms("geary", g, vals, idx), rtol=RTOL, atol=ATOL)
        plan.close()
        graph.close()
    assert not np.allclose(out[0], out[1], rtol=1e-9, atol=0)


@pytest.mark.parametrize("case", ["uniform", "classes", "classes-isolated-spot", "classes-balanced", "general"])
def test_geary_row_sum_shortcuts(L, ctx, perm_kernel, case):
    """Geary's permutations need `sum_i z_i^2 r[idx_p(i)]`, r = the graph's row sums.  `transformation=True` (the reference's default,
    gr/_ppatterns.py:212-214) row-normalises the graph: float64 weights give ONE row sum (the term is a constant: Moran's kernel),
    float32 weights — what `spatial_neighbors` stores — one value per degree, and so does a binary graph: a class table instead of a
    third random LDS read, and when one value holds on 3 spots in 4 (a grid: all but its border) that value times sum z^2 plus short
    exception lists for the other spots.  All shortcuts are exact: the same scores as the general kernel
    (SQGR_AUTOCORR_ROWSUM_CLASSES=0) and as each other (SQGR_AUTOCORR_ROWSUM_EXCEPTIONS=0: the class table) to the rounding of a
    re-ordered sum, and as the oracle."""
    import os

    from sklearn.preprocessing import normalize

    rng = np.random.default_rng(8)
    if case == "uniform":
        xy = rng.random((9000, 2))
        g = normalize(knn_graph(xy, 6).astype(np.float64), norm="l1", axis=1)
    else:
        g = O.hex_grid_graph(90, 100)                      # degrees 2..6, float32 weights
        xy = O.hex_grid(90, 100) / 9000.0
        if case == "general":
            g = g.astype(np.float64)
            g.data = rng.random(g.nnz) + 0.1
        else:
            g = normalize(g, norm="l1", axis=1)
            assert g.dtype == np.float32
            if case == "classes-isolated-spot":
                g = g.tolil()
                g[17, :] = 0
                g = g.tocsr()
                g.eliminate_zeros()
            if case == "classes-balanced":                 # every other row doubled: no row sum holds on 3 spots in 4
                g = sp.diags(np.where(np.arange(g.shape[0]) % 2 == 0, 1.0, 2.0).astype(np.float32)) @ g
                g = sp.csr_matrix(g).astype(np.float32)
    n, G, P = g.shape[0], 300, 70
    rs_vals, rs_counts = np.unique(np.asarray(g.astype(np.float64).sum(axis=1)).ravel(), return_counts=True)
    n_distinct, others = len(rs_vals), 1.0 - rs_counts.max() / n
    assert {"uniform": n_distinct == 1, "classes": 2 <= n_distinct <= 8 and others < 0.1, "classes-isolated-spot": 3 <= n_distinct <= 8 and others < 0.1,
            "classes-balanced": 2 <= n_distinct <= 8 and others > 0.25, "general": n_distinct > 8}[case]
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    vals[3] += 3 * np.sin(xy[:, 0] * 6)
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    fast = plan.perms("geary", seed=5, perm_begin=0, perm_end=P)
    others_plans = []
    for env in ("SQGR_AUTOCORR_ROWSUM_EXCEPTIONS", "SQGR_AUTOCORR_ROWSUM_CLASSES"):   # the class table; the general kernel
        os.environ[env] = "0"
        try:
            plan2 = L.AutocorrPlan(ctx, graph, vals)
            ref = plan2.perms("geary", seed=5, perm_begin=0, perm_end=P)
        finally:
            del os.environ[env]
        others_plans.append(plan2)
        np.testing.assert_allclose(fast, ref, rtol=1e-11, atol=1e-13)
        if case == "general" or (case in ("uniform", "classes-balanced") and env.endswith("EXCEPTIONS")):
            np.testing.assert_array_equal(fast, ref)  # the same kernel both times
    idx = np.stack([devrng.autocorr_permutation(n, 5, p) for p in (0, 1, 69)])
    np.testing.assert_allclose(fast[[0, 1, 69]], O.score_perms("geary", g, vals, idx), rtol=RTOL, atol=ATOL)
    # split invariance inside the shortcut: a range in two pieces, bit for bit
    two = np.concatenate([plan.perms("geary", seed=5, perm_begin=0, perm_end=33), plan.perms("geary", seed=5, perm_begin=33, perm_end=P)])
    np.testing.assert_array_equal(two, fast)
    # ... and numpy's streams / injected permutations take the same shortcut
    inj = plan.perms("geary", perm_idx=idx)
    np.testing.assert_allclose(inj, O.score_perms("geary", g, vals, idx), rtol=RTOL, atol=ATOL)
    plan.close()
    for p2 in others_plans:
        p2.close()
    graph.close()


def _adata(n=600, G=40, seed=0):
    import squidpy_amd as sq

    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2)) * 100
    X = rng.gamma(2.0, 1.0, size=(n, G))
    X[:, 1] += np.sin(xy[:, 0] / 10.0) * 2
    var = pd.DataFrame({"highly_variable": rng.random(G) < 0.5}, index=[f"gene{i}" for i in range(G)])
    obs = pd.DataFrame({"a": rng.random(n), "b": rng.integers(0, 5, n), "txt": ["x"] * n})
    return sq.AnnDataLite(
        X=X, obs=obs, var=var, obsm={"spatial": xy, "feat": rng.random((n, 3))},
        obsp={"spatial_connectivities": knn_graph(xy, 6)}, layers={"counts": X * 2.0},
    )


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_frontend_dataframe_equals_reference_pipeline(L, mode):
    """Whole-function parity of a DEFAULT call (the reference's permutation streams, reproduced on the device): every column of the result
    equals the oracle's restatement of gr/_ppatterns.py:196-255."""
    import squidpy_amd as sq

    adata = _adata()
    df = sq.gr.spatial_autocorr(adata, mode=mode, n_perms=50, seed=11, copy=True)  # the default: numpy's streams on the device
    hv = adata.var["highly_variable"].to_numpy()
    ref = O.spatial_autocorr(adata.obsp["spatial_connectivities"], adata.X[:, hv].T, adata.var_names[hv], mode=mode, n_perms=50, seed=11)
    stat = "I" if mode == "moran" else "C"
    assert list(df.columns) == list(ref.columns) and len(df.columns) == 9
    assert list(df.index) == list(ref.index)  # same sort order
    for c in df.columns:
        np.testing.assert_allclose(df[c].to_numpy(), ref[c].to_numpy(), rtol=RTOL, atol=ATOL, err_msg=c)
    assert f"pval_sim_fdr_bh" in df.columns and stat in df.columns


def test_frontend_structure_ported_from_reference_tests(L):
    """reference tests/graph/test_ppatterns.py:18-166,210-218."""
    import squidpy_amd as sq

    adata = _adata()
    assert sq.gr.spatial_autocorr(adata, mode="moran") is None and sq.gr.spatial_autocorr(adata, mode="geary") is None
    assert "moranI" in adata.uns and "gearyC" in adata.uns
    df = adata.uns["moranI"]
    assert df.shape[1] == 4 and "pval_norm_fdr_bh" in df.columns
    assert sorted(df.index) == sorted(adata.var_names[adata.var["highly_variable"]])
    assert (np.diff(df["I"].to_numpy()) <= 0).all() and (np.diff(adata.uns["gearyC"]["C"].to_numpy()) >= 0).all()
    # var_norm equals the closed forms (reference test_spatial_autocorr_var_norm_formula)
    from sklearn.preprocessing import normalize

    g = normalize(adata.obsp["spatial_connectivities"].copy(), norm="l1", axis=1)
    n = g.shape[0]
    s0, s1, s2 = O.g_moments(g)
    v_moran = (n * n * s1 - n * s2 + 3 * s0 * s0) / ((n - 1) * (n + 1) * s0 * s0) - (1.0 / (n - 1)) ** 2
    v_geary = ((2 * s1 + s2) * (n - 1) - 4 * s0 * s0) / (2 * (n + 1) * s0 * s0)
    np.testing.assert_allclose(df["var_norm"].to_numpy(), v_moran, rtol=1e-10)
    np.testing.assert_allclose(adata.uns["gearyC"]["var_norm"].to_numpy(), v_geary, rtol=1e-10)
    # reproducibility / seed: same seed same frame, different seed different sims
    a = sq.gr.spatial_autocorr(adata, n_perms=30, seed=1, copy=True, n_jobs=2, backend="threading")
    b = sq.gr.spatial_autocorr(adata, n_perms=30, seed=1, copy=True)
    c = sq.gr.spatial_autocorr(adata, n_perms=30, seed=2, copy=True)
    pd.testing.assert_frame_equal(a, b)
    assert a.shape[1] == 9 and not np.allclose(a["var_sim"], c.loc[a.index, "var_sim"])
    np.testing.assert_array_equal(a["I"], c.loc[a.index, "I"])
    # attr plumbing
    d_obs = sq.gr.spatial_autocorr(adata, attr="obs", copy=True)
    assert sorted(d_obs.index) == ["a", "b"] and np.isfinite(d_obs["I"]).all()
    d_obsm = sq.gr.spatial_autocorr(adata, attr="obsm", layer="feat", genes=[0, 2], copy=True)
    assert sorted(d_obsm.index) == [0, 2]
    d_layer = sq.gr.spatial_autocorr(adata, genes=["gene1", "gene3"], layer="counts", copy=True)
    d_x = sq.gr.spatial_autocorr(adata, genes=["gene1", "gene3"], copy=True)
    np.testing.assert_allclose(d_layer.loc[d_x.index, "I"], d_x["I"], rtol=1e-9)  # I is scale invariant
    d_one = sq.gr.spatial_autocorr(adata, genes="gene1", copy=True, corr_method=None)
    assert d_one.shape == (1, 3)
    # gene blocks and sparse X give the same numbers
    ad2 = adata.copy()
    ad2.X = sp.csr_matrix(ad2.X)
    e = sq.gr.spatial_autocorr(ad2, n_perms=30, seed=1, copy=True, gene_block=7)
    pd.testing.assert_frame_equal(a, e, rtol=1e-12)
    with pytest.raises(ValueError, match="Invalid option `foo` for `SpatialAutocorr`"):
        sq.gr.spatial_autocorr(adata, mode="foo")
    with pytest.raises(KeyError, match="not found in `adata.obsp`"):
        sq.gr.spatial_autocorr(adata, connectivity_key="nope_connectivities")
    with pytest.raises(ValueError, match="n_perms"):
        sq.gr.spatial_autocorr(adata, n_perms=0)


def test_statistical_properties_under_permutation(L, ctx):
    """Size-independent properties: E[I_perm] ~ -1/(N-1), E[C_perm] ~ 1; a spatially smooth feature is significant."""
    rng = np.random.default_rng(2)
    n, G = 5000, 64
    xy = rng.random((n, 2))
    from sklearn.preprocessing import normalize

    g = normalize(knn_graph(xy, 6), norm="l1", axis=1)
    vals = rng.normal(size=(G, n))
    vals[0] = np.sin(xy[:, 0] * 9) + 0.1 * rng.normal(size=n)
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    si = plan.perms("moran", seed=4, perm_begin=0, perm_end=200)
    sc = plan.perms("geary", seed=4, perm_begin=0, perm_end=200)
    assert abs(si.mean() + 1.0 / (n - 1)) < 5e-4 and abs(sc.mean() - 1.0) < 2e-3
    assert plan.scores("moran")[0] > si[:, 0].max() + 0.3 and plan.scores("geary")[0] < sc[:, 0].min() - 0.3


def test_numpy_device_streams_equal_host_streams(L):
    """rng="numpy" (PCG64 permutations generated on the GPU) gives exactly the frame of rng="numpy-host"."""
    import squidpy_amd as sq

    adata = _adata(n=400, G=24)
    a = sq.gr.spatial_autocorr(adata, mode="geary", n_perms=40, seed=5, copy=True, rng="numpy")
    b = sq.gr.spatial_autocorr(adata, mode="geary", n_perms=40, seed=5, copy=True, rng="numpy-host")
    pd.testing.assert_frame_equal(a, b)


def test_use_raw_and_missing_raw(L):
    """reference tests/graph/test_ppatterns.py:210-218 (`use_raw=True` reads `adata.raw`)."""
    import squidpy_amd as sq

    adata = _adata(n=300, G=12)
    with pytest.raises(AttributeError, match="No `.raw` attribute found"):
        sq.gr.spatial_autocorr(adata, use_raw=True, genes=["gene1"])
    adata.raw = sq.AnnDataLite(X=adata.X * 3.0, obs=adata.obs, var=adata.var)
    df_raw = sq.gr.spatial_autocorr(adata, use_raw=True, genes=["gene1", "gene2", "nope"], copy=True)
    df = sq.gr.spatial_autocorr(adata, genes=["gene1", "gene2"], copy=True)
    assert sorted(df_raw.index) == ["gene1", "gene2"]  # intersected with raw.var_names
    np.testing.assert_allclose(df_raw.loc[df.index, "I"], df["I"], rtol=1e-9)


def test_moran_geary_exact_rational_known_answers_on_the_device(L, ctx):
    """The HIP kernels against the exact-rational known answers of tests/golden/autocorr_kat.json (the reference's 5-node
    fixture graph, raw and row-normalised, and its 49-spot Visium fixture; constant features -> NaN).  Weights travel as
    float64, so agreement is to rounding (1e-12), far inside the 1e-6 bar."""
    from tests.test_oracle_pinned import _autocorr_kat

    for name, g, X, want in _autocorr_kat():
        graph = L.Graph(ctx, g)
        plan = L.AutocorrPlan(ctx, graph, X)
        np.testing.assert_allclose(plan.scores("moran"), want["I"], rtol=1e-12, atol=1e-15, equal_nan=True, err_msg=name)
        np.testing.assert_allclose(plan.scores("geary"), want["C"], rtol=1e-12, atol=1e-15, equal_nan=True, err_msg=name)
        # the permutation scores under the identity permutation are the observed scores
        idx = np.arange(g.shape[0], dtype=np.int32)[None, :]
        np.testing.assert_allclose(plan.perms("moran", perm_idx=idx)[0], want["I"], rtol=1e-12, atol=1e-15, equal_nan=True)
        np.testing.assert_allclose(plan.perms("geary", perm_idx=idx)[0], want["C"], rtol=1e-12, atol=1e-15, equal_nan=True)
        plan.close()
        graph.close()


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_permutation_reductions_on_the_device_are_numpys(L, ctx, mode):
    """`sqgr_autocorr_perm_stats` keeps the (P, G) scores on the device and returns what gr/_ppatterns.py:474-492 takes
    out of them: the exceedance counts and numpy's sum / std / var over the permutation axis — bit for bit numpy's own
    results on the scores of `perms()`, for all three permutation sources, incl. a constant (NaN) feature."""
    from squidpy_amd._utils import pcg64_states

    rng = np.random.default_rng(17)
    n, G, P = 1500, 70, 97
    g = knn_graph(rng.random((n, 2)), 6)
    g.data = rng.random(g.nnz).astype(np.float32) + 0.1
    vals = rng.gamma(2.0, 1.0, size=(G, n))
    vals[3] = 0.5
    graph = L.Graph(ctx, g)
    plan = L.AutocorrPlan(ctx, graph, vals)
    score = plan.scores(mode)
    states = pcg64_states(5, P)
    idx = O.autocorr_perm_indices(n, 8, P)
    for kw, sims in (
        (dict(seed=3, perm_begin=4, perm_end=4 + P), plan.perms(mode, seed=3, perm_begin=4, perm_end=4 + P)),
        (dict(pcg_states=states), plan.perms_pcg64(mode, states)),
        (dict(perm_idx=idx), plan.perms(mode, perm_idx=idx)),
    ):
        red = plan.perm_stats(mode, score, **kw)
        with np.errstate(invalid="ignore"):
            np.testing.assert_array_equal(red["n_ge"], (sims >= score).sum(axis=0))
            np.testing.assert_array_equal(red["sum"], sims.sum(axis=0))
            np.testing.assert_array_equal(red["std"], sims.std(axis=0))
            np.testing.assert_array_equal(red["var"], np.var(sims, axis=0))
        assert np.isnan(red["sum"][3]) and red["n_ge"][3] == 0
    plan.close()
    graph.close()


@pytest.mark.parametrize("P", [5, 8, 39, 128, 129, 1000, 8192, 8200, 20000])
def test_single_feature_reductions_follow_numpys_contiguous_order(L, ctx, P):
    """One feature in the whole call (`genes="x"`): the reference's score array is (P, 1), contiguous along the permutation
    axis, and numpy reduces it pairwise in 8192-element runs instead of row by row (it differs from the sequential sum from
    ~30 permutations on).  `only_feature=1` follows that order — bit for bit — and the front end sets it exactly when the call
    has one feature; a one-feature BLOCK of a larger call keeps the row-by-row order."""
    import squidpy_amd as sq

    rng = np.random.default_rng(P)
    n = 400
    g = knn_graph(rng.random((n, 2)), 6)
    g.data = rng.random(g.nnz) + 0.1
    graph = L.Graph(ctx, g)
    vals = rng.gamma(2.0, 1.0, size=(1, n))
    plan = L.AutocorrPlan(ctx, graph, vals)
    for mode in ("moran", "geary"):
        score = plan.scores(mode)
        sims = plan.perms(mode, seed=9, perm_begin=0, perm_end=P)
        assert sims.shape == (P, 1) and sims.flags.c_contiguous
        red = plan.perm_stats(mode, score, seed=9, perm_begin=0, perm_end=P, only_feature=True)
        np.testing.assert_array_equal(red["n_ge"], (sims >= score).sum(axis=0))
        np.testing.assert_array_equal(red["sum"], sims.sum(axis=0))
        np.testing.assert_array_equal(red["std"], sims.std(axis=0))
        np.testing.assert_array_equal(red["var"], np.var(sims, axis=0))
        wide = np.hstack([sims, sims])  # the same column inside a (P, 2) array: row by row
        red = plan.perm_stats(mode, score, seed=9, perm_begin=0, perm_end=P)
        np.testing.assert_array_equal(red["sum"], wide.sum(axis=0)[:1])
        np.testing.assert_array_equal(red["std"], wide.std(axis=0)[:1])
    plan.close()
    with pytest.raises(L.SqgrError, match="only_feature"):
        two = L.AutocorrPlan(ctx, graph, np.vstack([vals, vals]))
        try:
            two.perm_stats("moran", two.scores("moran"), seed=1, perm_begin=0, perm_end=4, only_feature=True)
        finally:
            two.close()
    graph.close()
    if P == 1000:  # the front end sets the flag exactly when the call has one feature
        from squidpy_amd.gr import _ppatterns as pp

        seen = []
        real = pp.AutocorrPlan.perm_stats

        def spy(self, *a, **kw):
            seen.append((self.G, kw.get("only_feature")))
            return real(self, *a, **kw)

        adata = _adata(n=300, G=12, seed=2)
        try:
            pp.AutocorrPlan.perm_stats = spy
            sq.gr.spatial_autocorr(adata, genes="gene1", n_perms=40, seed=3, copy=True)
            sq.gr.spatial_autocorr(adata, genes=["gene1", "gene2", "gene3"], n_perms=40, seed=3, copy=True, gene_block=2)
        finally:
            pp.AutocorrPlan.perm_stats = real
        assert seen == [(1, True), (2, False), (1, False)]


@pytest.mark.parametrize("fmt,dtype,itype", [("csr", np.float32, np.int32), ("csc", np.float32, np.int64), ("csr", np.float64, np.int64),
                                               ("csc", np.float64, np.int32), ("dense", np.float32, None), ("coo", np.int32, None)])
def test_sparse_and_float32_expression_resident_on_the_device(L, fmt, dtype, itype):
    """`adata.X` as real objects hold it — scipy CSR / CSC, float32, int32 or int64 index arrays; also dense float32 and an
    integer COO matrix — is uploaded as it is and densified / widened on the device (`sqgr_matrix_create_csr/_csc/_dense`):
    the result frame is IDENTICAL (every bit) to the one of the dense float64 matrix with the same values, for all features
    and for a feature subset, over several feature blocks."""
    import squidpy_amd as sq

    adata = _adata(n=700, G=53, seed=4)
    rng = np.random.default_rng(9)
    X = np.where(rng.random(adata.X.shape) < 0.15, np.round(adata.X * 4), 0.0)  # sparse counts, exactly representable in float32
    X[:, 7] = 0.0  # an all-zero (constant) feature
    dense = adata.copy()
    dense.X = X.astype(np.float64)
    if fmt == "dense":
        other_X = X.astype(dtype)
    else:
        m = getattr(sp, fmt + "_matrix")(X.astype(dtype))
        if itype is not None:
            m = type(m)((m.data, m.indices.astype(itype), m.indptr.astype(itype)), shape=m.shape)
        other_X = m
    other = adata.copy()
    other.X = other_X
    with pytest.warns(UserWarning, match="constant"):
        a = sq.gr.spatial_autocorr(dense, genes=list(dense.var_names), n_perms=40, seed=2, copy=True, gene_block=16)
    with pytest.warns(UserWarning, match="constant"):
        b = sq.gr.spatial_autocorr(other, genes=list(other.var_names), n_perms=40, seed=2, copy=True, gene_block=16)
    pd.testing.assert_frame_equal(a, b, check_exact=True)
    sub = ["gene40", "gene3", "gene12", "gene13"]
    pd.testing.assert_frame_equal(sq.gr.spatial_autocorr(dense, genes=sub, n_perms=20, seed=2, copy=True, mode="geary"),
                                  sq.gr.spatial_autocorr(other, genes=sub, n_perms=20, seed=2, copy=True, mode="geary"), check_exact=True)


def test_sparse_matrix_argument_checks(L, ctx):
    """The library checks scipy's canonical format where the arrays land (no host pass for the usual canonical matrix); a
    matrix with unsorted rows and repeated entries is repaired on the host (`sum_duplicates`, float32 sums like `toarray()`)
    and gives the frame of its dense form; an index outside the matrix is an error."""
    import squidpy_amd as sq

    m = sp.random(50, 20, density=0.2, format="csr", random_state=1, dtype=np.float32)
    dm = L.DeviceMatrix(ctx, m)
    assert dm.kind == "csr" and dm.shape == (50, 20)
    dm.close()
    adata = _adata(n=300, G=12, seed=6)
    rng = np.random.default_rng(1)
    rows = np.repeat(np.arange(300), 5)
    cols = rng.integers(0, 12, rows.size)  # unsorted inside the rows, with repeats
    vals = rng.integers(1, 9, rows.size).astype(np.float32) / 3
    indptr = np.arange(0, rows.size + 1, 5)
    messy = sp.csr_matrix((vals, cols, indptr), shape=(300, 12))
    dense = adata.copy()
    dense.X = messy.toarray().astype(np.float64)
    other = adata.copy()
    other.X = sp.csr_matrix((vals.copy(), cols.copy(), indptr.copy()), shape=(300, 12))
    pd.testing.assert_frame_equal(sq.gr.spatial_autocorr(dense, genes=list(dense.var_names), n_perms=16, seed=1, copy=True),
                                  sq.gr.spatial_autocorr(other, genes=list(other.var_names), n_perms=16, seed=1, copy=True), check_exact=True)
    bad = sp.csr_matrix((vals[:5], np.array([0, 3, 5, 7, 25]), np.array([0, 5] + [5] * 299)), shape=(300, 12), copy=True)
    with pytest.raises(L.SqgrError, match="outside"):
        L.DeviceMatrix(ctx, bad)


@pytest.mark.parametrize("fmt,dtype", [("csr", np.float32), ("csr", np.float64), ("csc", np.float32), ("dense", np.float64), ("dense", np.float32)])
def test_gene_subsets_are_selected_on_the_device(L, ctx, fmt, dtype, monkeypatch):
    """The reference's default (the highly variable genes) and explicit `genes` lists are `adata[:, genes].X` on the host
    (gr/_ppatterns.py:156-166); here the matrix is uploaded whole and `sqgr_autocorr_create_colidx` picks the columns on the
    device (a CSR matrix gets a by-column twin there): frames identical to the host-subset path, for any order, repeats,
    `use_raw`, over several feature blocks; a column outside the matrix is an error."""
    import squidpy_amd as sq
    from squidpy_amd.gr import _ppatterns as pp

    adata = _adata(n=900, G=61, seed=8)
    rng = np.random.default_rng(3)
    X = np.where(rng.random(adata.X.shape) < 0.2, np.round(adata.X * 4), 0.0)
    host = adata.copy()
    host.X = X.astype(np.float64)
    dev = adata.copy()
    dev.X = X.astype(dtype) if fmt == "dense" else getattr(sp, fmt + "_matrix")(X.astype(dtype))
    dev.raw = dev.copy()
    host.raw = host.copy()
    calls = []
    real = L.AutocorrPlan.from_column_list.__func__
    monkeypatch.setattr(pp.AutocorrPlan, "from_column_list", classmethod(lambda cls, *a: (calls.append(len(a[3])), real(cls, *a))[1]))
    monkeypatch.setattr(pp._ColumnSelection, "worthwhile", staticmethod(lambda base, cols: base is not host.X and base is not host.raw.X))
    picks = [list(adata.var_names[::-1][:33]), ["gene5", "gene60", "gene5", "gene0"], None]
    for genes in picks:
        calls.clear()
        a = sq.gr.spatial_autocorr(host, genes=genes, n_perms=24, seed=5, copy=True, gene_block=8)
        assert not calls
        b = sq.gr.spatial_autocorr(dev, genes=genes, n_perms=24, seed=5, copy=True, gene_block=8)
        assert calls and sum(calls) == len(b)
        pd.testing.assert_frame_equal(a, b, check_exact=True)
    calls.clear()
    a = sq.gr.spatial_autocorr(host, genes=["gene9", "gene1", "nope"], use_raw=True, mode="geary", copy=True)
    b = sq.gr.spatial_autocorr(dev, genes=["gene9", "gene1", "nope"], use_raw=True, mode="geary", copy=True)
    assert calls
    pd.testing.assert_frame_equal(a.sort_index(), b.sort_index(), check_exact=True)  # `set` order decides ties of the sort
    monkeypatch.undo()
    # the library call itself
    g = sp.csr_matrix(adata.obsp["spatial_connectivities"]).astype(np.float64)
    graph = L.Graph(ctx, g, with_data=True)
    dm = L.DeviceMatrix(ctx, dev.X)
    cols = np.array([60, 0, 17, 17, 33], dtype=np.int32)
    p1 = L.AutocorrPlan.from_column_list(ctx, graph, dm, cols)
    p2 = L.AutocorrPlan(ctx, graph, np.ascontiguousarray(X[:, cols].T))
    np.testing.assert_array_equal(p1.scores("moran"), p2.scores("moran"))
    np.testing.assert_array_equal(p1.scores("geary"), p2.scores("geary"))
    p1.close(), p2.close()
    with pytest.raises(L.SqgrError, match="outside"):
        L.AutocorrPlan.from_column_list(ctx, graph, dm, np.array([3, 61], dtype=np.int32))
    with pytest.raises(L.SqgrError, match="outside"):
        L.AutocorrPlan.from_column_list(ctx, graph, dm, np.array([-1], dtype=np.int32))
    dm.close()
    graph.close()


def test_matrix_buffers_are_parked_and_reused_without_stale_contents(L, ctx):
    """The payload buffers of a destroyed `sqgr_matrix` are parked for the next one (a 16 GB hipMalloc was seen to stall for
    seconds on a fragmented heap).  A reused buffer carries the last owner's bytes: every format must overwrite what it reads —
    a smaller matrix after a larger one, dense after sparse and back, give the results of a fresh process."""
    rng = np.random.default_rng(0)
    n, G = 6000, 4000  # 192 MB of float64: above the pool's 64 MB threshold
    g = knn_graph(rng.random((n, 2)), 6)
    g.data = rng.random(g.nnz) + 0.1
    graph = L.Graph(ctx, g)
    big = rng.gamma(2.0, 1.0, size=(n, G))
    small = rng.gamma(2.0, 1.0, size=(n, G - 900))  # fits the parked buffer (within 1.5x)
    sparse_x = sp.random(n, G, density=0.5, format="csr", random_state=3, dtype=np.float64)  # 96 MB of values (parked), 48 MB of indices
    want = {}
    for name, x in (("big", big), ("small", small), ("sparse", sparse_x)):
        dense = np.asarray(x.todense()) if sp.issparse(x) else x
        ref = L.AutocorrPlan(ctx, graph, np.ascontiguousarray(dense[:, 5:70].T))
        want[name] = ref.scores("moran")
        ref.close()
    for name, x in (("big", big), ("small", small), ("sparse", sparse_x), ("big", big), ("sparse", sparse_x), ("small", small)):
        dm = L.DeviceMatrix(ctx, x)
        plan = L.AutocorrPlan.from_columns(ctx, graph, dm, 5, 65)
        np.testing.assert_array_equal(plan.scores("moran"), want[name])
        plan.close()
        cols = np.arange(69, 4, -1, dtype=np.int32)
        plan = L.AutocorrPlan.from_column_list(ctx, graph, dm, cols)  # (CSR: the by-column twin comes out of the pool as well)
        np.testing.assert_array_equal(plan.scores("moran"), want[name][::-1])
        plan.close()
        dm.close()
    graph.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_matrix_streamed_in_column_blocks_gives_the_same_frame(L, ctx, dtype, monkeypatch):
    """Round 6: a large dense `adata.X` is uploaded column block by column block on the copy stream, by a second host thread, while
    the first feature blocks are scored (`DeviceMatrix(stream_columns=...)`, `sqgr_matrix_alloc_dense` +
    `sqgr_matrix_upload_columns`).  The frame must equal the one of the matrix uploaded whole in front, bit for bit — also for a
    column range through the row pitch (a wider host matrix), with a last block that is shorter than the others; the library's
    deferred `hipFree`s are handed to the driver afterwards (the allocator's counters say so)."""
    import squidpy_amd as sq

    rng = np.random.default_rng(2)
    n, G = 9000, 2300
    wide = rng.gamma(2.0, 1.0, size=(n, G + 40)).astype(dtype)
    X = wide[:, 17 : 17 + G]  # row pitch > row length
    adata = sq.AnnDataLite(X=X, obs=pd.DataFrame(index=[f"s{i}" for i in range(n)]), obsp={"spatial_connectivities": knn_graph(rng.random((n, 2)), 6)})
    kw = dict(mode="moran", n_perms=40, seed=3, copy=True, gene_block=256, show_progress_bar=False)
    monkeypatch.setattr(L.DeviceMatrix, "STREAM_MIN_BYTES", 1 << 62)
    whole = sq.gr.spatial_autocorr(adata, **kw)
    monkeypatch.setattr(L.DeviceMatrix, "STREAM_MIN_BYTES", 1 << 20)
    seen = []
    real_upload = ctx.lib.sqgr_matrix_upload_columns

    def counting_upload(*a):
        seen.append(int(a[4].value) if hasattr(a[4], "value") else int(a[4]))
        return real_upload(*a)

    monkeypatch.setattr(ctx.lib, "sqgr_matrix_upload_columns", counting_upload)
    a0 = ctx.alloc_counters()
    streamed = sq.gr.spatial_autocorr(adata, **kw)
    assert sum(seen) == G and len(seen) >= 9, "the matrix did not arrive in column blocks"
    pd.testing.assert_frame_equal(streamed, whole, check_exact=True)
    # a single-block DeviceMatrix directly: wait_columns + a column LIST (waits for every column)
    dm = L.DeviceMatrix(ctx, X, stream_columns=300)
    g = L.cached_graph(ctx, adata.obsp["spatial_connectivities"], with_data=True)
    plan = L.AutocorrPlan.from_column_list(ctx, g, dm, np.arange(G - 1, G - 200, -1, dtype=np.int32))
    ref = L.AutocorrPlan(ctx, g, np.ascontiguousarray(X[:, G - 1 : G - 200 : -1].T.astype(np.float64)))
    np.testing.assert_array_equal(plan.scores("moran"), ref.scores("moran"))
    plan.close()
    ref.close()
    dm.close()
    big = L.DeviceMatrix(ctx, np.ones((64, 64)))  # any allocation + release after the upload: the deferred frees go out
    big.close()
    a1 = ctx.alloc_counters()
    assert a1["frees"] - a0["frees"] >= a1["mallocs"] - a0["mallocs"] - 64, (a0, a1)
