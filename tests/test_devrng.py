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


# ------------------------------------------------------------------------------------------- two-level label generator
# perm_p = sigma_p o pi_{p // 16}: the 16 permutations of a group share the 8-round bijection and differ by a keyed 2-round
# network (sqgr_rng.h).  Per-permutation statistics must be those of the 8-round generator; the permutations of one group
# must look independent in everything the permutation tests use.


@pytest.mark.parametrize("n", [1, 2, 16, 17, 257, 1000, 100003])
def test_label_permutations_are_bijections_keyed_by_the_global_index(n):
    perms = np.array([0, 1, 15, 16, 17, 4095, 2**33 + 5])
    pis = D.label_permutations(n, 11, perms)
    for pi in pis:
        assert np.array_equal(np.sort(pi), np.arange(n))
    # a permutation depends on (seed, global index, library) only — not on which others are generated with it
    for j, p in enumerate(perms):
        assert np.array_equal(D.label_permutations(n, 11, np.array([p]))[0], pis[j])
    if n > 16:
        assert not np.array_equal(pis[0], pis[1]) and not np.array_equal(pis[2], pis[3])  # inside a group / across groups
        assert not np.array_equal(pis[0], D.label_permutations(n, 12, perms[:1])[0])
        assert not np.array_equal(pis[0], D.label_permutations(n, 11, perms[:1], lib=1)[0])


@pytest.mark.parametrize("n", [4, 5, 7])
def test_label_permutations_uniform_over_all_permutations_small_n(n):
    P = 60000
    pis = D.label_permutations(n, 321, np.arange(P))
    codes = (pis * (n ** np.arange(n))).sum(1)
    _, cnt = np.unique(codes, return_counts=True)
    nf = math.factorial(n)
    exp = P / nf
    chi = ((cnt - exp) ** 2 / exp).sum() + (nf - len(cnt)) * exp
    assert (chi - (nf - 1)) / math.sqrt(2 * (nf - 1)) < 4.5
    # and so are the permutations at a FIXED position of the group (every 16th), and pairs inside a group look independent:
    # pi_a o pi_b^-1 is again uniform over S_n
    rel = np.empty((P // 16, n), dtype=np.int64)
    g = pis.reshape(P // 16, 16, n)
    inv = np.argsort(g[:, 3], axis=1)
    rel = np.take_along_axis(g[:, 9], inv, axis=1)
    _, cnt = np.unique((rel * (n ** np.arange(n))).sum(1), return_counts=True)
    exp = len(rel) / nf
    chi = ((cnt - exp) ** 2 / exp).sum() + (nf - len(cnt)) * exp
    assert (chi - (nf - 1)) / math.sqrt(2 * (nf - 1)) < 4.5


@pytest.mark.parametrize("n", [49, 1000])
def test_label_permutations_position_and_adjacency_uniformity(n):
    P = 6000 if n == 49 else 1500
    pis = D.label_permutations(n, 7, np.arange(P))
    M = np.zeros((n, n))
    for i in range(n):
        M[i] = np.bincount(pis[:, i], minlength=n)
    exp = P / n
    chi = ((M - exp) ** 2 / exp).sum()
    dof = (n - 1) ** 2
    assert abs((chi - dof) / math.sqrt(2 * dof)) < 4.5
    diffs = np.bincount(((pis[:, 1:] - pis[:, :-1]) % n).ravel(), minlength=n).astype(float)
    e = diffs[1:].sum() / (n - 1)
    chi2 = ((diffs[1:] - e) ** 2 / e).sum()
    assert abs((chi2 - (n - 2)) / math.sqrt(2 * (n - 2))) < 4.5 and diffs[0] == 0


def _difference_z(n: int, pis: np.ndarray, lag: int, g: int = 1000) -> float:
    """chi-square z of the distribution of (pi(x + lag) - pi(x)) mod n over g bins, with the EXACT number of residues per bin
    as the expectation (n is not a multiple of g in general; a flat expectation alone gives z ~ 5 for a perfect generator)."""
    size = np.bincount(np.arange(1, n) * g // n, minlength=g).astype(float)
    d = ((pis[:, lag:] - pis[:, :-lag]) % n).ravel()
    cnt = np.bincount(d * g // n, minlength=g).astype(float)
    e = cnt.sum() * size / size.sum()
    return float((((cnt - e) ** 2 / e).sum() - (g - 1)) / math.sqrt(2 * (g - 1)))


@pytest.mark.parametrize("n", [100000, 174592])
def test_label_permutations_large_domain_difference_statistics(n):
    """The statistic that exposes too few rounds / a too weak round function first (see above), for the two-level form and
    for the independent 8-round form.  n = 174592 = 512 x 341 is the worst case for the low-digit round function: F_B is
    uniform on [0, 512) and reduced mod B = 341 ~ 2/3 * 512, so half of the residues are twice as likely as the others in
    EVERY low-digit round (the non-uniformity VERDICT r1 asked a test for): the images must still be uniform."""
    P = 64
    _, B, _ = D.domain_dims(n)
    for pis in (D.label_permutations(n, 99, np.arange(P)), D.permutation_batch(n, D.round_keys(98, np.arange(P)))):
        for lag in (1, B):
            z = _difference_z(n, pis, lag)
            assert abs(z) < 4.5, (n, lag, z)
    if n == 174592:
        assert D.domain_dims(n) == (512, 341, 511)
        # images of one fixed rank over many permutations: uniform over the LOW digit (where F_B's bias would sit) and the high one
        img = D.label_permutations(n, 5, np.arange(16 * 4000), x=np.array([777]))[:, 0]
        for dig, m in ((img % 341, 341), (img // 341, 512)):
            cnt = np.bincount(dig, minlength=m).astype(float)
            e = cnt.sum() / m
            z = (((cnt - e) ** 2 / e).sum() - (m - 1)) / math.sqrt(2 * (m - 1))
            assert abs(z) < 4.5, (m, z)


def test_same_group_labelings_have_independent_contingency_tables():
    """The permutations of a group enter the moments of a permutation test only through the contingency table J of their
    labelings: for independent arrangements J[a, b] = p_a p_b up to sampling noise.  2e5 spots, 30 labels:
    chi-square of the 120 same-group pairs of one group against the 256 cross-group pairs of two groups."""
    n, K = 200_000, 30
    lab_sorted = np.sort(np.random.default_rng(0).integers(0, K, n))
    L = lab_sorted[D.label_permutations(n, 3, np.arange(32))]
    pk = np.bincount(lab_sorted, minlength=K) / n
    E = np.outer(pk, pk) * n

    def chi(i, j):
        J = np.zeros((K, K))
        np.add.at(J, (L[i], L[j]), 1)
        return ((J - E) ** 2 / E).sum(), np.abs(np.diag(J) / n - pk * pk).max() / (pk * (1 - pk)).min()

    dof = (K - 1) ** 2
    same = np.array([chi(i, j) for i in range(16) for j in range(i + 1, 16)])
    cross = np.array([chi(i, j) for i in range(0, 16, 2) for j in range(16, 32, 2)])
    assert abs(cross[:, 0].mean() / dof - 1) < 0.05
    # same-group tables are a little over-dispersed (the 2-round network has B distinct high-digit shifts): bounded, and
    # what matters — the label correlation (J_aa - p_a^2) / (p_a (1 - p_a)) that scales the covariance of two permutations'
    # counts — stays below 1 %
    assert same[:, 0].mean() / dof < 1.25 and same[:, 0].max() / dof < 2.5, (same[:, 0].mean() / dof, same[:, 0].max() / dof)
    assert same[:, 1].max() < 0.01, same[:, 1].max()


def test_permutation_test_moments_and_group_covariance_vs_numpy_streams():
    """Counts on a small graph: moments agree with numpy's shuffles AND the counts of two permutations of one group are
    uncorrelated (16 * Var(group mean) == Var, average within-group correlation ~ 0)."""
    rows, cols, k, G = 30, 40, 4, 400
    adj = O.hex_grid_graph(rows, cols)
    labels = np.random.default_rng(0).integers(0, k, rows * cols).astype(np.uint32)
    P = 16 * G
    dev = O.nhood_perm_counts_philox(adj.indices, adj.indptr, labels, k, 5, 0, P).reshape(P, -1).astype(float)
    ref = O.nhood_perm_counts_numpy(adj.indices, adj.indptr, labels, k, 5, P).reshape(P, -1).astype(float)
    se = ref.std(0) / math.sqrt(P)
    assert (np.abs(dev.mean(0) - ref.mean(0)) < 5 * se * math.sqrt(2)).all()
    assert (np.abs(dev.var(0) / ref.var(0) - 1) < 6 * math.sqrt(2.0 / P) * 1.5).all()
    z = ((dev - dev.mean(0)) / dev.std(0)).reshape(G, 16, -1)
    s, ss = z.sum(1), (z**2).sum(1)
    within = ((s**2 - ss) / (16 * 15)).mean(0)  # average correlation of two distinct permutations of a group, per cell
    assert np.abs(within).max() < 5.0 / math.sqrt(G * 120), np.abs(within).max()


def test_independent_label_permutations_are_bijections_of_their_own():
    """The restatement of the generator's independent variant (sqgr_nhood.hip: k_shuffle_indep): every row a permutation of [0, n),
    keyed by the permutation index alone — rows of one 16-group share nothing, and differ from the two-level generator's."""
    n = 1234
    perms = np.arange(16, 48)
    pi = D.independent_label_permutations(n, 5, perms)
    assert pi.shape == (32, n)
    for row in pi:
        assert np.array_equal(np.sort(row), np.arange(n))
    assert len({row.tobytes() for row in pi}) == 32
    np.testing.assert_array_equal(pi[3], D.permutation(n, D.round_keys(5, np.array([19]))[0]))
    assert not np.array_equal(pi, D.label_permutations(n, 5, perms))
