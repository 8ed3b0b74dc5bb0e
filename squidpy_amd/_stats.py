"""Host-side O(P*G) statistics of ``spatial_autocorr``: analytic moments, permutation p-values and
multiple-testing correction, restated from the reference (gr/_ppatterns.py:443-559) and from
``statsmodels.stats.multitest.multipletests`` (third-party, absent in this image; call site gr/_ppatterns.py:239-245)."""

from __future__ import annotations

from typing import Any

import numpy as np
from scipy import sparse, stats


def g_moments(w: Any) -> tuple[float, float, float]:
    """gr/_ppatterns.py:541-559 (pysal's s0, s1, s2)."""
    s0 = w.sum()
    t = w.transpose() + w
    t2 = t.multiply(t) if sparse.issparse(t) else t * t
    s1 = t2.sum() / 2.0
    s2array = np.array(w.sum(1) + w.sum(0).transpose()) ** 2
    s2 = s2array.sum()
    return s0, s1, s2


def analytic_pval(score: np.ndarray, g: Any, mode: str, expected: float, two_tailed: bool) -> tuple[np.ndarray, float]:
    """gr/_ppatterns.py:501-538: Moran and Geary have different normality variances (Cliff & Ord 1981)."""
    s0, s1, s2 = g_moments(g)
    n = g.shape[0]
    s02 = s0 * s0
    if mode == "geary":
        v_norm = ((2 * s1 + s2) * (n - 1) - 4 * s02) / (2 * (n + 1) * s02)
    elif mode == "moran":
        n2 = n * n
        v_num = n2 * s1 - n * s2 + 3 * s02
        v_den = (n - 1) * (n + 1) * s02
        v_norm = v_num / v_den - (1.0 / (n - 1)) ** 2
    else:
        raise AssertionError(f"Unexpected mode `{mode}`.")
    se_norm = v_norm ** (1 / 2.0)
    z_norm = (score - expected) / se_norm
    p_norm = np.empty(score.shape)
    p_norm[z_norm > 0] = 1 - stats.norm.cdf(z_norm[z_norm > 0])
    p_norm[z_norm <= 0] = stats.norm.cdf(z_norm[z_norm <= 0])
    if two_tailed:
        p_norm *= 2.0
    return p_norm, v_norm


def p_value_calc(score: np.ndarray, sims: np.ndarray | None, g: Any, mode: str, expected: float, two_tailed: bool) -> dict[str, Any]:
    """gr/_ppatterns.py:443-498."""
    p_norm, var_norm = analytic_pval(score, g, mode, expected, two_tailed)
    results: dict[str, Any] = {"pval_norm": p_norm, "var_norm": var_norm}
    if sims is None:
        return results
    n_perms = sims.shape[0]
    large_perm = (sims >= score).sum(axis=0)
    sel = (n_perms - large_perm) < large_perm
    large_perm[sel] = n_perms - large_perm[sel]
    p_sim = (large_perm + 1) / (n_perms + 1)
    e_score_sim = sims.sum(axis=0) / n_perms
    se_score_sim = sims.std(axis=0)
    z_sim = (score - e_score_sim) / se_score_sim
    p_z_sim = np.empty(z_sim.shape)
    p_z_sim[z_sim > 0] = 1 - stats.norm.cdf(z_sim[z_sim > 0])
    p_z_sim[z_sim <= 0] = stats.norm.cdf(z_sim[z_sim <= 0])
    results["pval_z_sim"] = p_z_sim
    results["pval_sim"] = p_sim
    results["var_sim"] = np.var(sims, axis=0)
    return results


def _ecdf_adjust(ps: np.ndarray, factor: np.ndarray) -> np.ndarray:
    raw = ps / factor
    corrected = np.minimum.accumulate(raw[::-1])[::-1]
    corrected[corrected > 1] = 1
    return corrected


def multipletests_pvals(pvals: np.ndarray, method: str = "fdr_bh") -> np.ndarray:
    """Adjusted p-values of ``statsmodels.stats.multitest.multipletests(pvals, alpha=0.05, method=method)[1]``."""
    pvals = np.asarray(pvals, dtype=float)
    n = len(pvals)
    order = np.argsort(pvals)
    ps = pvals[order]
    m = method.lower()
    if m in ("bonferroni", "b"):
        corr = np.minimum(ps * float(n), 1.0)  # statsmodels clips at 1 at the end
    elif m in ("sidak", "s"):
        corr = -np.expm1(n * np.log1p(-ps))
    elif m in ("holm", "h"):
        corr = np.maximum.accumulate(ps * np.arange(n, 0, -1))
    elif m in ("holm-sidak", "hs"):
        corr = np.maximum.accumulate(-np.expm1(np.arange(n, 0, -1) * np.log1p(-ps)))
    elif m in ("simes-hochberg", "sh"):
        corr = np.minimum.accumulate((ps * np.arange(n, 0, -1))[::-1])[::-1]
    elif m in ("fdr_bh", "fdr_i", "fdr_p", "fdri", "fdrp"):
        corr = _ecdf_adjust(ps, np.arange(1, n + 1) / float(n))
    elif m in ("fdr_by", "fdr_n", "fdr_c", "fdrn", "fdrcorr"):
        cm = np.sum(1.0 / np.arange(1, n + 1))
        corr = _ecdf_adjust(ps, np.arange(1, n + 1) / float(n) / cm)
    else:
        raise ValueError(f"multiple-testing method `{method}` is not implemented in squidpy_amd")
    corr = np.asarray(corr, dtype=float)
    corr[corr > 1] = 1
    out = np.empty_like(corr)
    out[order] = corr
    return out
