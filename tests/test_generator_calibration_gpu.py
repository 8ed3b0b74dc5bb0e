"""Calibration of the DEFAULT permutation stream where it is weakest — one group (task 6).

The device generator (csrc/sqgr_rng.h) lets the 16 permutations of a group share one strong 8-round bijection and tells them
apart by a keyed 2-round network; with ``n_perms <= 16`` every permutation of a call comes from ONE group.  This test puts the
front ends through exactly that regime and compares against the mode that IS Squidpy (``rng="numpy"``: numpy's own streams
reproduced on the device): ``nhood_enrichment`` on BASELINE config 1's graph (5 000-spot hex grid, 10 clusters) and
``spatial_autocorr`` (``pval_sim``), ``n_perms`` in {16, 32, 100}, 200 seeds each.  For the two streams the per-cell mean and
variance of the z-score over the seeds and the pooled z distribution (two-sample Kolmogorov-Smirnov distance) must agree within
sampling error; the same for the Monte-Carlo p-values of Moran's I."""

import warnings

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

SEEDS = 200


def _ks(a: np.ndarray, b: np.ndarray) -> float:
    a, b = np.sort(a.ravel()), np.sort(b.ravel())
    grid = np.concatenate([a, b])
    return float(np.abs(np.searchsorted(a, grid, side="right") / a.size - np.searchsorted(b, grid, side="right") / b.size).max())


@pytest.fixture(scope="module")
def config1():
    import squidpy_amd as sq
    from squidpy_amd._synthetic import hex_grid_graph

    adj = hex_grid_graph(50, 100)
    n = adj.shape[0]
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 10, n)
    obs = pd.DataFrame({"cl": pd.Categorical.from_codes(labels, [f"c{i}" for i in range(10)])}, index=[f"s{i}" for i in range(n)])
    # 24 features: half spatially structured (so that small p-values occur), half noise
    xy = np.stack([np.arange(n) % 100, np.arange(n) // 100], 1).astype(np.float64)
    X = rng.gamma(2.0, 1.0, size=(n, 24))
    X[:, :12] += 0.15 * np.sin(xy[:, :1] / 7.0) * np.arange(1, 13)
    return sq, sq.AnnDataLite(X=X, obs=obs, obsp={"spatial_connectivities": adj})


@pytest.mark.parametrize("n_perms", [16, 32, 100])
def test_nhood_zscores_philox_vs_numpy_streams(config1, n_perms):
    sq, adata = config1
    z = {}
    for rng in ("philox", "numpy"):
        z[rng] = np.stack([sq.gr.nhood_enrichment(adata, "cl", n_perms=n_perms, seed=s, copy=True, rng=rng).zscore for s in range(SEEDS)])
        assert np.isfinite(z[rng]).all()
    a, b = z["philox"].reshape(SEEDS, -1), z["numpy"].reshape(SEEDS, -1)
    # per cell: means within sampling error of each other (100 cells: |N(0,1)| max ~ 2.8), variances likewise
    se_mean = np.sqrt((a.var(0, ddof=1) + b.var(0, ddof=1)) / SEEDS)
    z_mean = (a.mean(0) - b.mean(0)) / se_mean
    assert np.abs(z_mean).max() < 4.8, np.abs(z_mean).max()
    assert 0.6 < np.sqrt((z_mean**2).mean()) < 1.4, np.sqrt((z_mean**2).mean())
    # z = (count - mean_p) / std_p is t-like with n_perms - 1 degrees of freedom: excess kurtosis 6 / (dof - 4)
    kurt = 6.0 / max(n_perms - 5, 1)
    se_logvar = np.sqrt(2.0 * (2.0 + kurt) / SEEDS)
    log_ratio = np.log(a.var(0, ddof=1) / b.var(0, ddof=1))
    assert np.abs(log_ratio).max() < 5.0 * se_logvar, (np.abs(log_ratio).max(), se_logvar)
    assert abs(log_ratio.mean()) < 5.0 * se_logvar / np.sqrt(a.shape[1] / 4.0), log_ratio.mean()  # pooled over ~25 effective cells
    # pooled distribution of the z-scores (cells of one call are correlated: the bound is that of ~SEEDS * 25 independent values)
    d = _ks(a, b)
    assert d < 1.63 * np.sqrt(2.0 / (SEEDS * 25.0)), d
    # the first two pooled moments
    assert abs(a.mean() - b.mean()) < 0.05 and abs(np.log(a.std() / b.std())) < 0.04, (a.mean(), b.mean(), a.std(), b.std())


@pytest.mark.parametrize("n_perms", [16, 32, 100])
def test_moran_pval_sim_philox_vs_numpy_streams(config1, n_perms):
    sq, adata = config1
    p, zs = {}, {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rng in ("philox", "numpy"):
            frames = [sq.gr.spatial_autocorr(adata, mode="moran", n_perms=n_perms, seed=s, copy=True, rng=rng).sort_index() for s in range(SEEDS)]
            p[rng] = np.stack([f["pval_sim"].to_numpy() for f in frames])
            zs[rng] = np.stack([f["var_sim"].to_numpy() for f in frames])
    a, b = p["philox"], p["numpy"]
    assert ((a > 0) & (a <= 1)).all() and ((b > 0) & (b <= 1)).all()
    # per feature: mean Monte-Carlo p-value within sampling error (24 features)
    se = np.sqrt((a.var(0, ddof=1) + b.var(0, ddof=1)) / SEEDS) + 1e-12
    z_mean = (a.mean(0) - b.mean(0)) / se
    assert np.abs(z_mean).max() < 4.5, np.abs(z_mean).max()
    # pooled p-value distribution of the 12 noise features (the structured ones sit at the floor 1 / (n_perms + 1) in both)
    d = _ks(a[:, 12:], b[:, 12:])
    assert d < 1.63 * np.sqrt(2.0 / (SEEDS * 12.0)) + 0.5 / (n_perms + 1), d  # (+ half a step of the discrete p-value lattice)
    # variance of the permutation scores: the estimator the normal-approximation p-value divides by
    lr = np.log(zs["philox"].mean(0) / zs["numpy"].mean(0))
    assert np.abs(lr).max() < 5.0 * np.sqrt(2.0 / (SEEDS * (n_perms - 1)) * 2.0), np.abs(lr).max()
