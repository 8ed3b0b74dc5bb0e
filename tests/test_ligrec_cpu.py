"""CPU tests of the ligrec row: the oracle against the golden vectors of the reference's own
source, against the literal source where /root/reference exists, and the host logic of ``PermutationTest`` (the
reference's validation behaviour, tests/graph/test_ligrec.py:26-152, 436-444).  No device compute here."""

from __future__ import annotations

import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from oracle import ref_shim
from oracle import restate as O
from squidpy_amd import AnnDataLite
from squidpy_amd.gr import PermutationTest, ligrec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ligrec_reference.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_oracle_reproduces_reference_golden(gold, tag):
    means, pvals = O.ligrec_analysis(
        gold[f"{tag}_data"], gold[f"{tag}_clusters"], gold[f"{tag}_interactions"], gold[f"{tag}_cpairs"],
        threshold=float(gold[f"{tag}_threshold"]), n_perms=int(gold[f"{tag}_n_perms"]), seed=int(gold[f"{tag}_seed"]),
    )
    np.testing.assert_array_equal(means, gold[f"{tag}_means"])
    np.testing.assert_array_equal(pvals, gold[f"{tag}_pvalues"])


def test_golden_case_c_is_the_reference_nan_layout(gold):
    """tests/graph/test_ligrec.py:446-: 11 of 12 cells are NaN, GENE2->GENE3 in A->B is tested and has p = 0."""
    p = gold["C_pvalues"]
    assert np.isnan(p).sum() == 11 and p[1, 1] == 0.0


@pytest.mark.skipif(not ref_shim.available(), reason="needs /root/reference (build container)")
def test_oracle_matches_literal_reference_source():
    rng = np.random.default_rng(3)
    n, g, k = 180, 10, 5
    data = (rng.random((n, g)) < 0.4) * rng.gamma(2.0, 1.0, (n, g))
    cl = rng.integers(0, k, n).astype(np.int32)
    inter = np.array([(i, j) for i in range(g) for j in range(g)], dtype=np.int32)[::4]
    cp = np.array([(a, b) for a in range(k) for b in range(k)], dtype=np.int32)[::2]
    df = pd.DataFrame(data, columns=list(range(g)))
    df["clusters"] = pd.Categorical(cl)
    ref = ref_shim.ligrec()["_analysis"](df, inter, cp, threshold=0.4, n_perms=40, seed=9, n_jobs=1, show_progress_bar=False)
    means, pvals = O.ligrec_analysis(data, cl, inter, cp, threshold=0.4, n_perms=40, seed=9)
    np.testing.assert_array_equal(means, ref.means)
    np.testing.assert_array_equal(pvals, ref.pvalues)
    assert np.isnan(pvals).any() and (~np.isnan(pvals)).any()


# --------------------------------------------------------------------------------------------------------- host logic
@pytest.fixture()
def adata():
    rng = np.random.default_rng(0)
    n, g = 60, 16
    x = sp.csr_matrix((rng.random((n, g)) < 0.5) * rng.gamma(2.0, 1.0, (n, g)))
    var = pd.DataFrame({"symbol": [f"Sym{i}" for i in range(g)]}, index=[f"Gene{i}" for i in range(g)])
    obs = pd.DataFrame({"leiden": pd.Categorical(rng.integers(0, 3, n).astype(str)), "plain": rng.random(n)})
    raw = AnnDataLite(X=x, obs=obs.copy(), var=var.copy())
    return AnnDataLite(X=x[:, :8], obs=obs, var=var.iloc[:8].copy(), raw=raw)


@pytest.fixture()
def pairs(adata):
    g = list(adata.raw.var_names[:5])
    return [(a, b) for a in g for b in g]


def test_not_adata():
    with pytest.raises(TypeError, match=r"Expected `adata` to be of type `anndata.AnnData`, found `NoneType`."):
        PermutationTest(None)


def test_no_raw(adata):
    adata.raw = None
    with pytest.raises(AttributeError, match=r"No `.raw` attribute"):
        PermutationTest(adata, use_raw=True)
    PermutationTest(adata, use_raw=False)


def test_raw_with_other_cell_count(adata):
    adata.raw = adata.raw[np.arange(10), :]
    with pytest.raises(ValueError, match=r"Expected `60` cells in `.raw` object, found `10`."):
        PermutationTest(adata)


def test_invalid_complex_policy(adata, pairs):
    with pytest.raises(ValueError, match=r"Invalid option `foobar` for `ComplexPolicy`."):
        PermutationTest(adata).prepare(pairs, complex_policy="foobar")


def test_invalid_interactions(adata, pairs):
    pt = PermutationTest(adata)
    with pytest.raises(TypeError, match=r"Expected either a `pandas.DataFrame`"):
        pt.prepare(42)
    with pytest.raises(KeyError, match=r"Column .*target.* is not in `interactions`."):
        pt.prepare({"source": ["a"], "foo": ["b"]})
    with pytest.raises(KeyError, match=r"Column .*source.* is not in `interactions`."):
        pt.prepare(pd.DataFrame(pairs, columns=["foo", "target"]))
    with pytest.raises(ValueError, match=r"Not all interactions are of length `2`."):
        pt.prepare([("a", "b"), ("c",), ("d", "e")])
    with pytest.raises(ValueError, match=r"No interactions were specified."):
        pt.prepare([])
    with pytest.raises(ValueError, match=r"After filtering by genes"):
        pt.prepare(["foo", "bar", "baz"])
    with pytest.raises(ValueError, match=r"The interactions are empty"):
        pt.prepare(pd.DataFrame({"source": [], "target": []}))


def test_interaction_spellings_agree(adata):
    g = list(adata.raw.var_names[:3])
    a = PermutationTest(adata).prepare(g).interactions  # all ordered pairs
    b = PermutationTest(adata).prepare([(s, t) for s in g for t in g]).interactions
    c = PermutationTest(adata).prepare(([s for s in g for _ in g], [t for _ in g for t in g])).interactions
    d = PermutationTest(adata).prepare({"source": [s for s in g for _ in g], "target": [t for _ in g for t in g]}).interactions
    for other in (b, c, d):
        np.testing.assert_array_equal(a[["source", "target"]].values, other[["source", "target"]].values)
    assert len(a) == 9


def test_all_genes_capitalized_and_duplicates_dropped(adata, pairs):
    pt = PermutationTest(adata).prepare(pairs + pairs[:3])
    vals = pt.interactions[["source", "target"]].values
    assert len(vals) == 25 and all(v == v.upper() for row in vals for v in row)
    assert repr(pt) == "<PermutationTest[n_interaction=25]>"


def test_none_source_target(adata):
    g = adata.raw.var_names
    pt = PermutationTest(adata).prepare({"source": [None, g[0]], "target": [None, g[1]]})
    assert len(pt.interactions) == 1


def test_metadata_columns_survive(adata, pairs):
    df = pd.DataFrame(pairs, columns=["source", "target"])
    df["metadata"] = "foo"
    pt = PermutationTest(adata).prepare(df)
    assert list(pt.interactions.columns) == ["source", "target", "metadata"]


def _complexes(g):
    return [(g[0], g[1]), (f"{g[2]}_{g[3]}", g[4]), (g[5], f"{g[6]}_{g[7]}"), (f"{g[8]}_{g[9]}", f"{g[10]}_{g[11]}"), (f"foo_{g[12]}_bar_baz", g[13])]


def test_complex_policy_min(adata):
    g = list(adata.raw.var_names)
    mean = np.asarray(adata.raw.X.mean(axis=0)).ravel()
    pt = PermutationTest(adata).prepare(_complexes(g), complex_policy="min")
    assert pt.interactions.shape == (5, 2)

    def lo(i, j):
        return g[i] if mean[i] <= mean[j] else g[j]

    np.testing.assert_array_equal(pt.interactions["source"], [x.upper() for x in (g[0], lo(2, 3), g[5], lo(8, 9), g[12])])
    np.testing.assert_array_equal(pt.interactions["target"], [x.upper() for x in (g[1], g[4], lo(6, 7), lo(10, 11), g[13])])


def test_complex_policy_all(adata):
    g = [x.upper() for x in adata.raw.var_names]
    pt = PermutationTest(adata).prepare(_complexes(list(adata.raw.var_names)), complex_policy="all")
    want = [(g[0], g[1]), (g[2], g[4]), (g[3], g[4]), (g[5], g[6]), (g[5], g[7]), (g[8], g[10]), (g[8], g[11]), (g[9], g[10]), (g[9], g[11]), (g[12], g[13])]
    assert [tuple(r) for r in pt.interactions[["source", "target"]].values] == want


def test_validation_before_any_device_work(adata, pairs):
    pt = PermutationTest(adata).prepare(pairs)
    with pytest.raises(KeyError, match=r"foobar"):
        pt.test("foobar")
    with pytest.raises(TypeError, match=r"categorical"):
        pt.test("plain")
    with pytest.raises(ValueError, match=r"Expected `n_perms` to be positive"):
        pt.test("leiden", n_perms=0)
    with pytest.raises(ValueError, match=r"Invalid option `foobar` for `CorrAxis`."):
        pt.test("leiden", corr_method="fdr_bh", corr_axis="foobar")
    with pytest.raises(ValueError, match=r"Invalid cluster `'foo'`."):
        pt.test("leiden", clusters=["foo"])
    with pytest.raises(ValueError, match=r"Expected a `tuple` of length `2`, found `3`."):
        pt.test("leiden", clusters=[("0", "1"), ("0", "1", "2")])
    with pytest.raises(ValueError, match=r"Number of threads must be"):
        pt.test("leiden", n_jobs=0)
    with pytest.raises(ValueError, match=r"Invalid option `mt` for `rng`."):
        pt.test("leiden", rng="mt")
    adata.obs["one"] = pd.Categorical(["a"] * adata.n_obs)
    with pytest.raises(ValueError, match=r"Expected at least `2` clusters, found `1`."):
        pt.test("one")


def test_deprecated_parallelization_params_warn(adata, pairs):
    pt = PermutationTest(adata).prepare(pairs)
    for param in ("numba_parallel", "backend"):
        with pytest.warns(FutureWarning, match=rf"Parameter `{param}` of `test\(\)` is deprecated"), pytest.raises(KeyError):
            pt.test("foobar", **{param: True})
        with pytest.warns(FutureWarning, match=rf"Parameter `{param}` of `ligrec\(\)` is deprecated"), pytest.raises(KeyError):
            ligrec(adata, "foobar", interactions=pairs, **{param: True})


def test_gene_symbols_are_swapped_in_and_restored(adata):
    names = adata.raw.var_names.copy()
    sym = list(adata.raw.var["symbol"][:3])
    with pytest.raises(KeyError, match=r"foobar"):  # reaches the cluster-key check: the symbols were accepted
        ligrec(adata, "foobar", interactions=sym, gene_symbols="symbol")
    np.testing.assert_array_equal(adata.raw.var_names, names)
    with pytest.raises(KeyError, match=r"Unable to find gene symbols in `adata.raw.var\['nope'\]`."):
        ligrec(adata, "leiden", interactions=sym, gene_symbols="nope")
    with pytest.raises(ValueError, match=r"After filtering by genes"):  # without the swap the symbols are unknown genes
        ligrec(adata, "leiden", interactions=sym)


def test_single_cluster_subset_is_computed_like_the_reference(adata, pairs):
    """`clusters=[("0", "0")]` resolves to ONE cluster: the reference's `_analysis` has no check and computes it
    (gr/_ligrec.py:677-775; only `<= 1` categories in `obs` are rejected, :300).  No device is needed: a constant label vector
    does not change under shuffling: every permuted mean is the kernel's sequential sum / size, which either exceeds pandas'
    observed mean by a rounding error in ALL permutations or in none (p = 1 or 0).  Checked against the literal reference source."""
    res = PermutationTest(adata).prepare(pairs).test("leiden", clusters=[("0", "0")], n_perms=7, seed=1, copy=True, threshold=0.2)
    assert res["pvalues"].shape == (len(res["means"]), 1)
    pv = res["pvalues"].to_numpy(dtype=np.float64)
    assert set(np.unique(pv[~np.isnan(pv)])) <= {0.0, 1.0}
    if ref_shim.available():
        genes = [g.upper() for g in dict.fromkeys(g for p in pairs for g in p)]
        x = adata.raw.X.toarray()[:, :5]
        keep = (adata.obs["leiden"].astype(str) == "0").to_numpy()
        df = pd.DataFrame(x[keep], columns=list(range(5)))
        df["clusters"] = pd.Categorical(np.zeros(int(keep.sum()), dtype=np.int32))
        gi = {g: i for i, g in enumerate(genes)}
        inter = np.array([(gi[a], gi[b]) for a, b in res["means"].index], dtype=np.int32)
        ref = ref_shim.ligrec()["_analysis"](df, inter, np.array([(0, 0)], dtype=np.int32), threshold=0.2, n_perms=7, seed=1, n_jobs=1,
                                           show_progress_bar=False)
        np.testing.assert_array_equal(res["means"].to_numpy(dtype=np.float64), np.asarray(ref.means))
        np.testing.assert_array_equal(pv, np.asarray(ref.pvalues))
