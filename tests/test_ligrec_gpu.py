"""GPU parity tests of the ligand-receptor permutation test: libsqgr (HIP) vs the CPU oracle
and vs golden vectors produced by the reference's own source (tests/golden/make_ligrec_golden.py).

Counts are integers and compared bit for bit.  Group means are float64 sums accumulated in the reference's order and
are compared bit for bit as well."""

from __future__ import annotations

import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from oracle import restate as O
from squidpy_amd import AnnDataLite
from squidpy_amd._utils import pcg64_states

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ligrec_reference.npz")


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


def _problem(n, g, k, seed, density=0.3, integer=False, n_inter=None, pairs=None):
    rng = np.random.default_rng(seed)
    data = (rng.random((n, g)) < density) * (rng.poisson(2.0, (n, g)) + 1.0 if integer else rng.gamma(2.0, 1.0, (n, g)))
    data = data.astype(np.float64)
    cl = rng.integers(0, k, n).astype(np.int32)
    cl[:k] = np.arange(k)  # every cluster is populated
    inter = np.array([(i, j) for i in range(g) for j in range(g)], dtype=np.int32)
    if n_inter is not None:
        inter = inter[rng.choice(len(inter), n_inter, replace=False)]
    cp = np.array([(a, b) for a in range(k) for b in range(k)], dtype=np.int32) if pairs is None else np.asarray(pairs, np.int32)
    return data, cl, inter, cp


def _device_counts(L, ctx, data, cl, k, inter, cp, pre, **kw):
    return L.ligrec_counts(ctx, sp.csc_matrix(data), cl, k, pre["inv_counts"], inter, cp, pre["obs"], pre["valid"].astype(np.uint8), **kw)


@pytest.mark.parametrize(
    "n,g,k,n_perms,integer",
    [(300, 12, 4, 64, False), (257, 7, 3, 100, True), (1000, 33, 20, 130, False), (64, 5, 2, 1, True), (500, 9, 100, 70, False)],
)
def test_counts_numpy_streams_bit_exact(L, ctx, n, g, k, n_perms, integer):
    data, cl, inter, cp = _problem(n, g, k, seed=n + g)
    if integer:
        data = np.rint(data)
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.05)
    labels = O.ligrec_perm_labels_numpy(cl, 11, n_perms)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got, groups = _device_counts(
        L, ctx, data, cl, k, inter, cp, pre, pcg_states=pcg64_states(11, n_perms), perm_begin=0, perm_end=n_perms, return_first_groups=True
    )
    np.testing.assert_array_equal(got, want)
    # the sums are accumulated in cell order on the device too: identical bits, not merely close
    np.testing.assert_array_equal(groups, O.ligrec_group_means(data, labels[0], pre["inv_counts"]))
    assert want.sum() > 0


def test_counts_device_generator_match_oracle(L, ctx):
    data, cl, inter, cp = _problem(400, 10, 5, seed=5, n_inter=40)
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.1)
    labels = O.ligrec_perm_labels_philox(cl, 1234, 3, 3 + 70)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got, groups = _device_counts(L, ctx, data, cl, 5, inter, cp, pre, seed=1234, perm_begin=3, perm_end=73, return_first_groups=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(groups, O.ligrec_group_means(data, labels[0], pre["inv_counts"]))


def test_more_than_256_clusters_in_tiles(L, ctx):
    """256 < K <= 2048: 16-bit labels from both generators, group sums in cluster tiles of <= 255 clusters sharing one
    permutation, scores straight from global memory.  Counts and group means bit for bit like the small-K path."""
    n, g, k = 1500, 6, 300
    data, cl, inter, cp = _problem(n, g, k, seed=21, density=0.5, n_inter=12)
    rng = np.random.default_rng(3)
    cp = cp[rng.choice(len(cp), 700, replace=False)]
    cp[:4] = [(0, 299), (299, 0), (254, 255), (150, 151)]  # pairs across and at the tile borders
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.0)
    # numpy streams
    labels = O.ligrec_perm_labels_numpy(cl, 5, 40)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got, groups = _device_counts(L, ctx, data, cl, k, inter, cp, pre, pcg_states=pcg64_states(5, 40), perm_begin=0, perm_end=40, return_first_groups=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(groups, O.ligrec_group_means(data, labels[0], pre["inv_counts"]))
    assert want.sum() > 0
    # device generator, a range that starts inside a group of 16, and its split
    labels = O.ligrec_perm_labels_philox(cl, 99, 5, 5 + 50)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got, groups = _device_counts(L, ctx, data, cl, k, inter, cp, pre, seed=99, perm_begin=5, perm_end=55, return_first_groups=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(groups, O.ligrec_group_means(data, labels[0], pre["inv_counts"]))
    parts = [_device_counts(L, ctx, data, cl, k, inter, cp, pre, seed=99, perm_begin=a, perm_end=b) for a, b in [(5, 21), (21, 55)]]
    np.testing.assert_array_equal(got, sum(parts))


@pytest.mark.parametrize("k", [3000, 40000])
def test_thousands_of_clusters(L, ctx, k):
    """More than 2048 clusters (the reference has no limit, gr/_ligrec.py:616-673): the shufflers address 16-bit labels; at
    40 000 clusters the label-boundary table no longer fits LDS next to the block table and the device generator's exact
    route reads it from global memory.  Counts and group means bit for bit like the oracle's, both generators."""
    n, g = 45000, 4
    data, cl, inter, cp = _problem(n, g, k, seed=77, density=0.6, n_inter=6, pairs=[(0, 1)])
    rng = np.random.default_rng(8)
    cp = np.stack([rng.integers(0, k, 400), rng.integers(0, k, 400)], axis=1).astype(np.int32)
    cp[:4] = [(0, k - 1), (k - 1, 0), (254, 255), (2048, 2049)]
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.0)
    labels = O.ligrec_perm_labels_numpy(cl, 11, 20)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got, groups = _device_counts(L, ctx, data, cl, k, inter, cp, pre, pcg_states=pcg64_states(11, 20), perm_begin=0, perm_end=20, return_first_groups=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(groups, O.ligrec_group_means(data, labels[0], pre["inv_counts"]))
    assert want.sum() > 0
    labels = O.ligrec_perm_labels_philox(cl, 5, 3, 3 + 20)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got, groups = _device_counts(L, ctx, data, cl, k, inter, cp, pre, seed=5, perm_begin=3, perm_end=23, return_first_groups=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(groups, O.ligrec_group_means(data, labels[0], pre["inv_counts"]))


def test_permutation_ranges_add_up(L, ctx):
    """Sharding invariance: counts over [0, P) equal the sum over any split (what the multi-GPU path relies on)."""
    data, cl, inter, cp = _problem(350, 8, 6, seed=9)
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.0)
    full = _device_counts(L, ctx, data, cl, 6, inter, cp, pre, seed=77, perm_begin=0, perm_end=200)
    parts = [_device_counts(L, ctx, data, cl, 6, inter, cp, pre, seed=77, perm_begin=a, perm_end=b) for a, b in [(0, 37), (37, 101), (101, 200)]]
    np.testing.assert_array_equal(full, sum(parts))
    st = pcg64_states(5, 200)
    full = _device_counts(L, ctx, data, cl, 6, inter, cp, pre, pcg_states=st, perm_begin=0, perm_end=200)
    parts = [_device_counts(L, ctx, data, cl, 6, inter, cp, pre, pcg_states=st[a:b], perm_begin=a, perm_end=b) for a, b in [(0, 64), (64, 65), (65, 200)]]
    np.testing.assert_array_equal(full, sum(parts))


def test_empty_columns_and_invalid_cells(L, ctx):
    data, cl, inter, cp = _problem(200, 6, 3, seed=2)
    data[:, 2] = 0.0  # a gene without any stored entry
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.2)
    assert not pre["valid"].all() and pre["valid"].any()
    labels = O.ligrec_perm_labels_numpy(cl, 3, 50)
    want = O.ligrec_score_permutations(data, labels, pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    got = _device_counts(L, ctx, data, cl, 3, inter, cp, pre, pcg_states=pcg64_states(3, 50), perm_begin=0, perm_end=50)
    np.testing.assert_array_equal(got, want)
    assert (got[~pre["valid"]] == 0).all()
    # zero permutations: all-zero counts
    got0 = _device_counts(L, ctx, data, cl, 3, inter, cp, pre, seed=1, perm_begin=5, perm_end=5)
    assert got0.shape == want.shape and not got0.any()


def test_bad_arguments_fail_loudly(L, ctx):
    data, cl, inter, cp = _problem(50, 4, 3, seed=1)
    pre = O.ligrec_prepare(data, cl, inter, cp, threshold=0.0)
    with pytest.raises(L.SqgrError, match="gene id"):
        _device_counts(L, ctx, data, cl, 3, inter + 10, cp, pre, seed=1, perm_begin=0, perm_end=4)
    with pytest.raises(L.SqgrError, match="cluster id"):
        _device_counts(L, ctx, data, cl, 3, inter, cp + 3, pre, seed=1, perm_begin=0, perm_end=4)
    with pytest.raises(L.SqgrError):
        _device_counts(L, ctx, data, cl + 5, 3, inter, cp, pre, seed=1, perm_begin=0, perm_end=4)


# ------------------------------------------------------------------------------------------------------------- front end
def _adata_from(data, cl, names=None):
    g = data.shape[1]
    var = pd.DataFrame(index=[f"G{i}" for i in range(g)] if names is None else names)
    obs = pd.DataFrame({"cluster": pd.Categorical([f"c{c}" for c in cl])})
    return AnnDataLite(X=sp.csr_matrix(data), obs=obs, var=var)


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_front_end_reproduces_reference_golden(gold, tag):
    """A default call (numpy's streams, reproduced on the device): p-values of the reference's own `_analysis` source for the same seed, exactly."""
    import squidpy_amd as sq

    data, cl = gold[f"{tag}_data"], gold[f"{tag}_clusters"]
    inter, cp = gold[f"{tag}_interactions"], gold[f"{tag}_cpairs"]
    adata = _adata_from(data, cl)
    res = sq.gr.ligrec(
        adata, "cluster", interactions=[(f"G{s}", f"G{t}") for s, t in inter], clusters=[(f"c{a}", f"c{b}") for a, b in cp],
        threshold=float(gold[f"{tag}_threshold"]), n_perms=int(gold[f"{tag}_n_perms"]), seed=int(gold[f"{tag}_seed"]),
        use_raw=False, copy=True,
    )
    assert list(res["pvalues"].index) == [(f"G{s}", f"G{t}") for s, t in inter]
    assert list(res["pvalues"].columns) == [(f"c{a}", f"c{b}") for a, b in cp]
    np.testing.assert_array_equal(res["means"].to_numpy(), gold[f"{tag}_means"])
    np.testing.assert_array_equal(res["pvalues"].to_numpy(dtype=np.float64), gold[f"{tag}_pvalues"])  # NaN == NaN here


def test_front_end_more_than_256_clusters():
    """The reference has no cluster limit (gr/_ligrec.py:677-775); 300 clusters through the front end, rng='numpy': the
    oracle's `_analysis` restatement for the same seed, exactly."""
    import squidpy_amd as sq

    k = 300
    data, cl, inter, _ = _problem(1200, 5, k, seed=8, density=0.6, n_inter=6)
    adata = _adata_from(data, cl)
    pairs = [(a, b) for a in (0, 1, 150, 254, 255, 256, 299) for b in (2, 255, 256, 298)]
    res = sq.gr.ligrec(
        adata, "cluster", interactions=[(f"G{s}", f"G{t}") for s, t in inter], clusters=[(f"c{a}", f"c{b}") for a, b in pairs],
        threshold=0.0, n_perms=30, seed=12, use_raw=False, copy=True, rng="numpy",
    )
    # the front end codes the clusters of the requested pairs in category order (strings: "c0", "c1", "c150", ...)
    cats = sorted({f"c{c}" for ab in pairs for c in ab})
    code = {c: i for i, c in enumerate(cats)}
    sel = np.isin(cl, [int(c[1:]) for c in cats])
    lab = np.array([code[f"c{c}"] for c in cl[sel]], dtype=np.int32)
    cp = np.array([(code[f"c{a}"], code[f"c{b}"]) for a, b in pairs], dtype=np.int32)
    assert len(cats) <= 256  # a subset: this call runs the small path; the full set below runs the tiles
    _, pv = O.ligrec_analysis(data[sel], lab, inter, cp, threshold=0.0, n_perms=30, seed=12)
    np.testing.assert_array_equal(res["pvalues"].to_numpy(dtype=np.float64), pv)
    # all 300 clusters (90 000 pairs)
    res = sq.gr.ligrec(adata, "cluster", interactions=[(f"G{s}", f"G{t}") for s, t in inter], threshold=0.0, n_perms=20, seed=4,
                       use_raw=False, copy=True, rng="numpy")
    cats = sorted(f"c{c}" for c in range(k))
    code = {c: i for i, c in enumerate(cats)}
    lab = np.array([code[f"c{c}"] for c in cl], dtype=np.int32)
    cp = np.array([(a, b) for a in range(k) for b in range(k)], dtype=np.int32)
    _, pv = O.ligrec_analysis(data, lab, inter, cp, threshold=0.0, n_perms=20, seed=4)
    assert res["pvalues"].shape == (len(inter), k * k)
    np.testing.assert_array_equal(res["pvalues"].to_numpy(dtype=np.float64), pv)


def test_front_end_device_generator_statistics():
    """rng='philox' follows another stream: same NaN pattern and means, p-values agree within Monte-Carlo error."""
    import squidpy_amd as sq

    data, cl, inter, cp = _problem(600, 10, 4, seed=21, n_inter=30)
    data[cl == 2, :3] *= 3.0
    adata = _adata_from(data, cl)
    kw = dict(interactions=[(f"G{s}", f"G{t}") for s, t in inter], threshold=0.05, n_perms=2000, use_raw=False, copy=True)
    a = sq.gr.ligrec(adata, "cluster", seed=1, rng="philox", **kw)
    b = sq.gr.ligrec(adata, "cluster", seed=1, rng="numpy", **kw)
    a2 = sq.gr.ligrec(adata, "cluster", seed=1, rng="philox", **kw)
    pa, pb = a["pvalues"].to_numpy(dtype=np.float64), b["pvalues"].to_numpy(dtype=np.float64)
    np.testing.assert_array_equal(pa, a2["pvalues"].to_numpy(dtype=np.float64))  # reproducible
    np.testing.assert_array_equal(np.isnan(pa), np.isnan(pb))
    np.testing.assert_array_equal(a["means"].to_numpy(), b["means"].to_numpy())
    ok = ~np.isnan(pa)
    # binomial standard error of a difference of two estimates at n_perms=2000 is <= sqrt(2*0.25/2000) = 0.0158
    assert np.abs(pa[ok] - pb[ok]).max() < 5 * 0.0158
    assert np.abs(pa[ok] - pb[ok]).mean() < 0.0158


def test_front_end_slots_fdr_and_clusters_subset():
    import squidpy_amd as sq

    data, cl, inter, cp = _problem(300, 8, 5, seed=4, n_inter=20)
    adata = _adata_from(data, cl)
    inter_names = [(f"g{s}", f"G{t}") for s, t in inter]  # lower case on purpose: genes are compared in upper case
    assert sq.gr.ligrec(adata, "cluster", interactions=inter_names, n_perms=30, seed=0, use_raw=False, clusters=["c0", "c3"]) is None
    res = adata.uns["cluster_ligrec"]
    assert set(res) == {"means", "pvalues", "metadata"}
    assert list(res["pvalues"].columns) == [("c0", "c0"), ("c0", "c3"), ("c3", "c0"), ("c3", "c3")]
    # the same through the oracle on the cell subset (numpy streams act on the SUBSET's label vector)
    keep = np.isin(cl, [0, 3])
    sub_cl = np.where(cl[keep] == 0, 0, 1).astype(np.int32)
    genes = sorted({f"G{i}" for pair in inter for i in pair})
    gid = {g: i for i, g in enumerate(genes)}
    sub = data[keep][:, [int(g[1:]) for g in genes]]
    ii = np.array([[gid[f"G{s}"], gid[f"G{t}"]] for s, t in inter], dtype=np.int32)
    cpp = np.array([(0, 0), (0, 1), (1, 0), (1, 1)], dtype=np.int32)
    res_np = sq.gr.ligrec(adata, "cluster", interactions=inter_names, n_perms=30, seed=0, use_raw=False, clusters=["c0", "c3"], copy=True, rng="numpy")
    means, pvals = O.ligrec_analysis(sub, sub_cl, ii, cpp, threshold=0.01, n_perms=30, seed=0)
    np.testing.assert_array_equal(res_np["means"].to_numpy(), means)
    np.testing.assert_array_equal(res_np["pvalues"].to_numpy(dtype=np.float64), pvals)
    # FDR along both axes keeps the NaN pattern and the [0, 1] range
    for axis in ("interactions", "clusters"):
        r = sq.gr.ligrec(adata, "cluster", interactions=inter_names, n_perms=30, seed=0, use_raw=False, copy=True, corr_method="fdr_bh", corr_axis=axis, key_added="foo")
        q = r["pvalues"].to_numpy(dtype=np.float64)
        assert np.nanmin(q) >= 0 and np.nanmax(q) <= 1
    assert "foo" not in adata.uns


def test_pvalues_reference_held_by_the_reference_repo():
    """The reference's own pinned result (tests/graph/test_ligrec.py:346-360 `test_pvalues_reference` against
    tests/_data/ligrec_pvalues_reference.h5ad): ligrec(adata, "leiden", interactions=product(raw.var_names[:5] x 2), n_perms=25,
    seed=42) on the tests/_data/test_data.h5ad fixture (raw = the AnnData itself, tests/conftest.py:40-41).  Both files are
    exported to tests/golden/*.npz by `make_golden.py --export-h5ad`; the DEFAULT call (numpy's streams on the device) must reproduce Squidpy's
    p-values for that seed: assert_allclose like the reference test, and the same NaN pattern."""
    import os
    from itertools import product

    import pandas as pd

    import squidpy_amd as sq

    gold = os.path.join(os.path.dirname(__file__), "golden")
    v = np.load(os.path.join(gold, "visium49.npz"))
    ref = np.load(os.path.join(gold, "ligrec_pvalues_reference.npz"))
    names = [str(s) for s in v["var_names40"]]
    obs = pd.DataFrame({"leiden": pd.Categorical.from_codes(v["leiden_codes"].astype(int), [str(c) for c in v["leiden_categories"]])},
                       index=[f"s{i}" for i in range(len(v["leiden_codes"]))])
    X = v["X40"].astype(np.float32)
    raw = sq.AnnDataLite(X=X, obs=obs.copy(), var=pd.DataFrame(index=names))
    adata = sq.AnnDataLite(X=X, obs=obs, var=pd.DataFrame(index=names), raw=raw)
    interactions = tuple(product(names[:5], names[:5]))
    r = sq.gr.ligrec(adata, "leiden", interactions=interactions, n_perms=25, copy=True, show_progress_bar=False, seed=42, n_jobs=1)  # the reference test's call, verbatim
    index = pd.MultiIndex.from_arrays([ref["source"], ref["target"]], names=["source", "target"])
    columns = pd.MultiIndex.from_arrays([ref["cluster_1"], ref["cluster_2"]], names=["cluster_1", "cluster_2"])
    for key in ("means", "pvalues"):
        np.testing.assert_array_equal(np.array(r[key].index.tolist()), np.array(index.tolist()))
        np.testing.assert_array_equal(np.array(r[key].columns.tolist()), np.array(columns.tolist()))
    np.testing.assert_allclose(r["means"].to_numpy(dtype=np.float64), ref["means"])
    np.testing.assert_allclose(r["pvalues"].to_numpy(dtype=np.float64), ref["pvalues"])
    np.testing.assert_array_equal(np.where(np.isnan(r["pvalues"].to_numpy(dtype=np.float64))), np.where(np.isnan(ref["pvalues"])))
