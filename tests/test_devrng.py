"""CPU tests of the device permutation generator's restatement (oracle/devrng.py == csrc/sqgr_rng.h, proven
equal on the GPU by tests/test_nhood_gpu.py::test_device_shuffle_matches_oracle_generator): Philox known answers,
bijectivity, statistical quality, and agreement of permutation-test moments with numpy's PCG64 shuffles."""

from __future__ import annotations

import math

import numpy as np
import pytest
from scipy import stats

from oracle import devrng as D
from oracle import restate as O


def test_philox4x32_10_known_answers():
    """Random123 known-answer vectors for Philox4x32-10."""
    z = D.philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(v) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = D.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(v) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = D.philox4x32_10(np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32), (0xA4093822, 0x299F31D0))[0]
    assert [hex(v) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 255, 256, 257, 1000, 65537, 100003])
def test_bijection_and_domain(n):
    A, B, Bmask = D.domain_dims(n)
    assert Bmask >= B - 1 and (Bmask + 1) & Bmask == 0 and Bmask < 2 * B
    assert A >= 16 and B >= 16 and A * B >= n and A < 2**16 and B < 2**16 and A & (A - 1) == 0
    if n > 512:
        assert A * B - n < A  # cycle walking almost never iterates
    for perm in (0, 3):
        pi = D.permutation(n, D.round_keys(11, np.array([perm]))[0])
        assert np.array_equal(np.sort(pi), np.arange(n))
    a = D.permutation(n, D.round_keys(11, np.array([0]))[0])
    b = D.permutation(n, D.round_keys(12, np.array([0]))[0])
    c = D.permutation(n, D.round_keys(11, np.array([0]), lib=1)[0])
    if n > 16:
        assert not np.array_equal(a, b) and not np.array_equal(a, c)


@pytest.mark.parametrize("n", [4, 5, 7])
def test_uniform_over_all_permutations_small_n(n):
    """chi-square over all n! permutations (P = 60 000 keys)."""
    P = 60000
    rks = D.round_keys(123, np.arange(P))
    pis = D.permutation_batch(n, rks)
    assert np.array_equal(pis[5], D.permutation(n, rks[5]))  # batch form == scalar form
    codes = (pis * (n ** np.arange(n))).sum(1)
    _, cnt = np.unique(codes, return_counts=True)
    nf = math.factorial(n)
    exp = P / nf
    chi = ((cnt - exp) ** 2 / exp).sum() + (nf - len(cnt)) * exp
    z = (chi - (nf - 1)) / math.sqrt(2 * (nf - 1))
    assert z < 4.5, (n, chi, z)


@pytest.mark.parametrize("n", [49, 1000])
def test_position_and_adjacency_uniformity(n):
    P = 6000 if n == 49 else 1500
    rks = D.round_keys(7, np.arange(P))
    pis = D.permutation_batch(n, rks)
    M = np.zeros((n, n))
    for i in range(n):
        M[i] = np.bincount(pis[:, i], minlength=n)
    diffs = np.bincount(((pis[:, 1:] - pis[:, :-1]) % n).ravel(), minlength=n).astype(float)
    exp = P / n
    chi = ((M - exp) ** 2 / exp).sum()
    dof = (n - 1) ** 2
    assert abs((chi - dof) / math.sqrt(2 * dof)) < 4.5
    e = diffs[1:].sum() / (n - 1)
    chi2 = ((diffs[1:] - e) ** 2 / e).sum()
    assert abs((chi2 - (n - 2)) / math.sqrt(2 * (n - 2))) < 4.5 and diffs[0] == 0


def test_permutation_test_moments_agree_with_numpy_streams():
    """Null distribution of neighbourhood counts under the device generator vs numpy's PCG64 shuffles."""
    rows, cols, k, P = 30, 40, 4, 600
    adj = O.hex_grid_graph(rows, cols)
    labels = np.random.default_rng(0).integers(0, k, rows * cols).astype(np.uint32)
    dev = O.nhood_perm_counts_philox(adj.indices, adj.indptr, labels, k, 5, 0, P)
    ref = O.nhood_perm_counts_numpy(adj.indices, adj.indptr, labels, k, 5, P)
    se = ref.std(0) / math.sqrt(P)
    assert (np.abs(dev.mean(0) - ref.mean(0)) < 5 * se * math.sqrt(2)).all()
    ratio = dev.var(0) / ref.var(0)
    assert (np.abs(ratio - 1) < 6 * math.sqrt(2.0 / P) * 1.5).all(), ratio
    for a in range(k):  # distributions of individual cells agree (two-sample KS)
        assert stats.ks_2samp(dev[:, a, a], ref[:, a, a]).pvalue > 1e-4


def test_library_shuffle_preserves_multisets():
    """reference tests/graph/test_utils.py:69-89 (`_shuffle_group`) for the device generator."""
    rng = np.random.default_rng(1)
    labels = rng.integers(0, 5, 500)
    libs = rng.integers(0, 3, 500)
    out = D.shuffled_labels(labels, 3, 9, libs, 3)
    for c in range(3):
        assert np.array_equal(np.sort(out[libs == c]), np.sort(labels[libs == c]))
    assert not np.array_equal(out, labels)


def test_large_domain_difference_statistics():
    """n = 1e5 (domain 512 x 196): images of neighbouring ranks (x, x+1: same high digit) and of ranks one low-digit
    period apart (x, x+B: same low digit) must differ by a uniformly distributed amount.  This is the statistic that
    exposes too few rounds / a too weak round function first (a 6-round variant with a half-range low-digit term scores
    z ~ 20 here while passing every small-n test)."""
    n, P, g = 100000, 120, 1000
    pis = D.permutation_batch(n, D.round_keys(99, np.arange(P)))
    _, B, _ = D.domain_dims(n)
    for lag in (1, B):
        d = ((pis[:, lag:] - pis[:, :-lag]) % n).ravel()
        cnt = np.bincount(d * g // n, minlength=g).astype(float)
        e = cnt.sum() / g
        z = (((cnt - e) ** 2 / e).sum() - (g - 1)) / math.sqrt(2 * (g - 1))
        assert abs(z) < 4.5, (lag, z)
