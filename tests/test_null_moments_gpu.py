"""Acceptance battery of the device label generator at FULL size (BASELINE config 5's graph: 1e6-spot hex grid, 30
clusters), against numpy's own shuffles reproduced on the device (rng="numpy", bit-identical to Squidpy's streams):

* mean and variance of every one of the 900 count cells over 102 400 permutations each (a bias of ~0.5 % of sigma in any
  cell fails) — the statistic the permutation test is about (promoted from tools/null_moments.py);
* the two-level structure of the generator (16 permutations share one strong bijection, csrc/sqgr_rng.h): counts of two
  permutations of the same group must be uncorrelated, and 16 * Var(group mean) == Var."""
import math

import numpy as np
import pytest

from oracle import restate as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


def test_null_moments_and_group_independence_on_the_config5_graph(L):
    from squidpy_amd._utils import pcg64_states

    ctx = L.default_context()
    P, k = 102_400, 30
    adj = O.hex_grid_graph(1000, 1000)
    n = adj.shape[0]
    labels = np.random.default_rng(0).integers(0, k, n).astype(np.int32)
    g = L.Graph(ctx, adj, with_data=False)
    plan = L.NhoodPlan(ctx, g, labels, k)
    assert plan.info()["symmetric"] and plan.info()["generator_group"] == 16
    freq = np.bincount(labels, minlength=k) / n
    shift = np.rint(adj.nnz * np.outer(freq, freq)).astype(np.int64)

    def moments(s1, s2):
        mean = s1.astype(np.float64) / P
        return mean, s2.astype(np.float64) / P - mean * mean

    s1, s2, perms = plan.run(20240925, 0, P, shift, return_perms=True)
    m_dev, v_dev = moments(s1, s2)
    # the moments returned by the device are those of the per-permutation counts
    d = perms.reshape(P, -1).astype(np.int64) - shift.reshape(-1)
    assert np.array_equal(d.sum(0), s1.reshape(-1)) and np.array_equal((d * d).sum(0).astype(np.uint64), s2.reshape(-1))
    s1n, s2n, _ = plan.run_pcg64(pcg64_states(7, P), shift)
    m_np, v_np = moments(s1n, s2n)
    z_mean = (m_dev - m_np) / np.sqrt((v_dev + v_np) / P)
    z_var = (v_dev - v_np) / (0.5 * (v_dev + v_np) * np.sqrt(4.0 / P))
    assert np.abs(z_mean).max() < 5.0 and np.abs(z_var).max() < 5.0, (np.abs(z_mean).max(), np.abs(z_var).max())
    assert 0.75 < np.sqrt((z_mean**2).mean()) < 1.25 and 0.75 < np.sqrt((z_var**2).mean()) < 1.25
    # ---- group structure: permutations 16q .. 16q+15 share the strong bijection
    G = P // 16
    x = d.astype(np.float64)
    z = ((x - x.mean(0)) / x.std(0)).reshape(G, 16, -1)
    s, ss = z.sum(1), (z**2).sum(1)
    within = ((s**2 - ss) / (16 * 15)).mean(0)             # average correlation of two permutations of one group, per cell
    se = 1.0 / math.sqrt(G * 120)
    assert np.abs(within).max() < 5.0 * se, (np.abs(within).max(), se)
    assert abs(within.mean()) < 5.0 * se / math.sqrt(30), within.mean()   # ~900 correlated cells: conservative
    ratio = z.mean(1).var(0) * 16                           # 16 * Var(group mean) / Var == 1 for independent permutations
    assert np.abs(ratio - 1).max() < 6.0 * math.sqrt(2.0 / G), np.abs(ratio - 1).max()
    # across groups (lag 16) nothing either
    lag = (z[:-1, 0] * z[1:, 0]).mean(0)
    assert np.abs(lag).max() < 5.0 / math.sqrt(G - 1)
    plan.close()
    g.close()
