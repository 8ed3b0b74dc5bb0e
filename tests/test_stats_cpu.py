"""CPU tests of the host-side statistics (p-values, multiple testing) against the golden outputs of the reference's
literal `_p_value_calc` / `_analytic_pval` / `_g_moments` and hand-computed corrections."""

from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

from squidpy_amd import _stats


def _g(golden):
    n = len(golden["autocorr_g_indptr"]) - 1
    return sp.csr_matrix((golden["autocorr_g_data"], golden["autocorr_g_indices"], golden["autocorr_g_indptr"]), shape=(n, n))


@pytest.mark.parametrize("mode", ["moran", "geary"])
def test_p_value_columns_match_reference_source(golden, mode):
    """`_stats` takes the permutation scores already reduced (the device forms the reductions, tests/test_autocorr_gpu.py
    checks those against numpy): fed with numpy's reductions of the golden scores it must give the columns of the
    reference's literal `_p_value_calc` / `_analytic_pval` / `_g_moments` (tests/golden/make_golden.py)."""
    g = _g(golden)
    n = g.shape[0]
    expected = -1.0 / (n - 1) if mode == "moran" else 1.0
    score, sims = golden[f"unpinned_{mode}_score"], golden[f"unpinned_{mode}_sims"]
    res = _stats.analytic_columns(score, g, mode, expected, False)
    res.update(_stats.permutation_columns(score, sims.shape[0], (sims >= score).sum(axis=0), sims.sum(axis=0), sims.std(axis=0), np.var(sims, axis=0)))
    for key in ("pval_norm", "pval_z_sim", "pval_sim", "var_sim"):
        np.testing.assert_allclose(res[key], golden[f"autocorr_{mode}_{key}"], rtol=1e-13)
    np.testing.assert_allclose(res["var_norm"], golden[f"autocorr_{mode}_var_norm"], rtol=1e-13)
    np.testing.assert_allclose(_stats.weight_moments(g), golden["autocorr_moments"], rtol=1e-13)
    np.testing.assert_allclose(_stats.weight_moments(g.toarray()), golden["autocorr_moments"], rtol=1e-6)  # dense float32 weights: other summation order
    two = _stats.analytic_columns(score, g, mode, expected, True)
    np.testing.assert_allclose(two["pval_norm"], 2 * golden[f"autocorr_{mode}_pval_norm"], rtol=1e-13)
    assert set(two) == {"pval_norm", "var_norm"}
    with pytest.raises(AssertionError, match="Unexpected mode"):
        _stats.variance_under_normality(_stats.weight_moments(g), n, "foo")


def test_multipletests_equals_statsmodels():
    """Pinned to statsmodels 0.12.2 itself (tests/golden/make_multipletests_golden.py, run with the image's conda python):
    every method squidpy_amd implements, on ties / zeros / ones / tiny values / a 400-vector / a NaN."""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multipletests_golden.json")) as fh:
        gold = json.load(fh)
    assert gold["statsmodels"] == "0.12.2"
    for name, case in gold["cases"].items():
        p = np.array(case["pvals"], dtype=float)
        for method, want in case["corrected"].items():
            want = np.array([np.nan if v is None else v for v in want], dtype=float)
            with np.errstate(invalid="ignore", divide="ignore"):
                got = _stats.multipletests_pvals(p, method)
            # Sidak / Holm-Sidak: statsmodels 0.12.2 evaluates `1 - (1 - p)**n`, which cancels (relative error ~ n * 1e-16 / p, exactly 0
            # for p below 1e-16); squidpy_amd keeps the `-expm1(n * log1p(-p))` form statsmodels switched to in 0.13 — the two agree to
            # the accuracy of the power form
            sidak = "sidak" in method
            # The power form `1 - (1 - p)**n` of 0.12.2 carries an ABSOLUTE error of a few ulps of 1 (the cancellation), whatever p: the
            # two forms are compared at rtol 1e-12 plus atol 1e-14 — ordinary p-values stay pinned to twelve digits (round
            # 4's blanket rtol 1e-6 hid them), the tiny ones, where the power form returns 0, agree to its 1e-14
            tol = dict(rtol=1e-12, atol=1e-14) if sidak else dict(rtol=1e-12 if name == "with_nan" else 1e-13, atol=0)
            # (with_nan: statsmodels sorts the NaN to the end and lets it poison what the method accumulates over it; the reference feeds
            # NaN p-values (constant features) straight in, so this behaviour is part of the contract)
            np.testing.assert_allclose(got, want, equal_nan=(name == "with_nan"), err_msg=f"{name}/{method}", **tol)


def test_multipletests_methods():
    p = np.array([0.01, 0.04, 0.03, 0.20, 0.5])
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "fdr_bh"), [0.05, 0.2 / 3, 0.2 / 3, 0.25, 0.5], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "bonferroni"), [0.05, 0.2, 0.15, 1.0, 1.0], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "holm"), [0.05, 0.12, 0.12, 0.4, 0.5], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "sidak"), 1 - (1 - p) ** 5, rtol=1e-12)
    tiny = np.array([1e-300, 1e-20, 1e-17, 0.5])  # where the power form of statsmodels 0.12 returns 0: n * p survives
    np.testing.assert_allclose(_stats.multipletests_pvals(tiny, "sidak")[:3], 4 * tiny[:3], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(tiny, "holm-sidak")[:3], [4e-300, 3e-20, 2e-17], rtol=1e-12)
    cm = sum(1 / k for k in range(1, 6))
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "fdr_by"), np.minimum(np.array([0.05, 0.2 / 3, 0.2 / 3, 0.25, 0.5]) * cm, 1), rtol=1e-12)
    with pytest.raises(ValueError, match="not implemented"):
        _stats.multipletests_pvals(p, "fdr_tsbky")
