"""Host-side arithmetic of ``spatial_autocorr``'s p-value columns, O(G) and O(nnz) only.

The O(P*G) part of the reference's ``_p_value_calc`` (gr/_ppatterns.py:474-492: the exceedance counts and the sum / std /
var of the permutation scores) is formed on the device next to the scores (``sqgr_autocorr_perm_stats``) and arrives
here already reduced; what is left is the normal tail, the folding of the counts, the closed-form variance under
normality (Cliff & Ord 1981, as pysal/esda and gr/_ppatterns.py:501-538 use it) and the multiple-testing correction
(``statsmodels.stats.multitest.multipletests``, third party; call site gr/_ppatterns.py:239-245) — pinned to statsmodels
0.12.2 itself by tests/golden/multipletests_golden.json."""

from __future__ import annotations

from typing import Any, NamedTuple

import numpy as np
from scipy import sparse, stats


class WeightMoments(NamedTuple):
    """pysal's S0, S1, S2 of a spatial weight matrix (gr/_ppatterns.py:541-559)."""

    s0: float
    s1: float
    s2: float


def weight_moments(w: Any) -> WeightMoments:
    """``S0 = sum w_ij``, ``S1 = 1/2 sum (w_ij + w_ji)^2``, ``S2 = sum_i (w_i. + w_.i)^2``."""
    sym = w.transpose() + w
    squared = sym.multiply(sym) if sparse.issparse(sym) else sym * sym
    # (row sums + column sums, kept in the shape and dtype the reference sums them in — (n, 1) for scipy matrices, whose
    # weights are often float32: the float32 pairwise sums depend on it in the 7th digit)
    margins = np.asarray(w.sum(axis=1) + w.sum(axis=0).transpose())
    return WeightMoments(w.sum(), squared.sum() / 2.0, np.square(margins).sum())


def variance_under_normality(mom: WeightMoments, n: int, mode: str) -> float:
    """Sampling variance of the statistic under the normality assumption.  Moran's I and Geary's C differ
    (gr/_ppatterns.py:513-526):  Var[I] = (n^2 S1 - n S2 + 3 S0^2) / ((n - 1)(n + 1) S0^2) - 1/(n - 1)^2,
    Var[C] = ((2 S1 + S2)(n - 1) - 4 S0^2) / (2 (n + 1) S0^2)."""
    s0sq = mom.s0 * mom.s0
    if mode == "moran":
        return (n * n * mom.s1 - n * mom.s2 + 3 * s0sq) / ((n - 1) * (n + 1) * s0sq) - (1.0 / (n - 1)) ** 2
    if mode == "geary":
        return ((2 * mom.s1 + mom.s2) * (n - 1) - 4 * s0sq) / (2 * (n + 1) * s0sq)
    raise AssertionError(f"Unexpected mode `{mode}`.")


def upper_or_lower_tail(z: np.ndarray) -> np.ndarray:
    """One-sided normal p-value of a z-score, folded at 0 the way the reference folds it: ``1 - cdf(z)`` where ``z > 0``,
    ``cdf(z)`` elsewhere (NaN stays NaN).  ``1 - cdf`` and not ``sf``: the reference's rounding is part of its result."""
    z = np.asarray(z, dtype=np.float64)
    cdf = stats.norm.cdf(z)
    return np.where(z > 0, 1 - cdf, cdf)


def analytic_columns(score: np.ndarray, w: Any, mode: str, expected: float, two_tailed: bool) -> dict[str, Any]:
    """``pval_norm`` / ``var_norm`` (gr/_ppatterns.py:501-538)."""
    var_norm = variance_under_normality(weight_moments(w), w.shape[0], mode)
    p = upper_or_lower_tail((score - expected) / var_norm ** (1 / 2.0))
    return {"pval_norm": p * 2.0 if two_tailed else p, "var_norm": var_norm}


def permutation_columns(score: np.ndarray, n_perms: int, n_ge: np.ndarray, sim_sum: np.ndarray, sim_std: np.ndarray, sim_var: np.ndarray) -> dict[str, Any]:
    """``pval_z_sim`` / ``pval_sim`` / ``var_sim`` (gr/_ppatterns.py:474-496) from the device's reductions of the permutation
    scores: ``n_ge`` = #{scores >= observed} is folded to the smaller tail, ``(k + 1) / (P + 1)``; the z-score of the
    observed statistic against the permutation mean ``sum / P`` and standard deviation goes through the normal tail."""
    k = np.minimum(np.asarray(n_ge, dtype=np.int64), n_perms - np.asarray(n_ge, dtype=np.int64))
    z = (score - sim_sum / n_perms) / sim_std
    return {"pval_z_sim": upper_or_lower_tail(z), "pval_sim": (k + 1) / (n_perms + 1), "var_sim": sim_var}


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
        # statsmodels >= 0.13 (`-np.expm1(ntests * np.log1p(-pvals))`): exact for tiny p, where the plain `1 - (1 - p)**n` of
        # 0.12.2 cancels to 0.  The reference asks for statsmodels >= 0.12 only; the form current installs evaluate is kept, the
        # 0.12.2 golden vectors are matched to the accuracy the power form has (tests/test_stats_cpu.py)
        corr = -np.expm1(n * np.log1p(-ps))
    elif m in ("holm", "h"):
        corr = np.maximum.accumulate(ps * np.arange(n, 0, -1))
    elif m in ("holm-sidak", "hs"):
        corr = np.maximum.accumulate(-np.expm1(np.arange(n, 0, -1) * np.log1p(-ps)))
    elif m in ("simes-hochberg", "sh"):
        corr = np.minimum.accumulate((ps * np.arange(n, 0, -1))[::-1])[::-1]
    elif m in ("hommel", "ho"):
        corr = ps.copy()
        for k in range(n, 1, -1):  # Hommel's step-up over the k largest p-values
            cim = np.min(k * ps[-k:] / np.arange(1, k + 1.0))
            corr[-k:] = np.maximum(corr[-k:], cim)
            corr[:-k] = np.maximum(corr[:-k], np.minimum(k * ps[:-k], cim))
    elif m == "fdr_gbs":
        ii = np.arange(1, n + 1)
        q = (n + 1.0 - ii) / ii * ps / (1.0 - ps)
        corr = np.minimum.accumulate(np.maximum.accumulate(q)[::-1])[::-1]
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
