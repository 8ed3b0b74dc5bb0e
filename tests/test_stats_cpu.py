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
def test_p_value_calc_matches_reference_source(golden, mode):
    g = _g(golden)
    n = g.shape[0]
    expected = -1.0 / (n - 1) if mode == "moran" else 1.0
    res = _stats.p_value_calc(golden[f"unpinned_{mode}_score"], golden[f"unpinned_{mode}_sims"], g, mode, expected, False)
    for key in ("pval_norm", "pval_z_sim", "pval_sim", "var_sim"):
        np.testing.assert_allclose(res[key], golden[f"autocorr_{mode}_{key}"], rtol=1e-13)
    np.testing.assert_allclose(res["var_norm"], golden[f"autocorr_{mode}_var_norm"], rtol=1e-13)
    np.testing.assert_allclose(_stats.g_moments(g), golden["autocorr_moments"], rtol=1e-13)
    two = _stats.p_value_calc(golden[f"unpinned_{mode}_score"], None, g, mode, expected, True)
    np.testing.assert_allclose(two["pval_norm"], 2 * golden[f"autocorr_{mode}_pval_norm"], rtol=1e-13)
    assert set(two) == {"pval_norm", "var_norm"}


def test_multipletests_methods():
    p = np.array([0.01, 0.04, 0.03, 0.20, 0.5])
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "fdr_bh"), [0.05, 0.2 / 3, 0.2 / 3, 0.25, 0.5], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "bonferroni"), [0.05, 0.2, 0.15, 1.0, 1.0], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "holm"), [0.05, 0.12, 0.12, 0.4, 0.5], rtol=1e-12)
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "sidak"), 1 - (1 - p) ** 5, rtol=1e-12)
    cm = sum(1 / k for k in range(1, 6))
    np.testing.assert_allclose(_stats.multipletests_pvals(p, "fdr_by"), np.minimum(np.array([0.05, 0.2 / 3, 0.2 / 3, 0.25, 0.5]) * cm, 1), rtol=1e-12)
    with pytest.raises(ValueError, match="not implemented"):
        _stats.multipletests_pvals(p, "fdr_tsbky")
