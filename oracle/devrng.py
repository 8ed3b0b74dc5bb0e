"""TEST INFRASTRUCTURE — not product code.

CPU (numpy) restatement of the *device* permutation generator used by the HIP path when
``rng="philox"`` (``squidpy_amd/csrc/sqgr_rng.h``): Philox4x32-10 derives eight 32-bit round
keys per (seed, permutation index, library), of which the low 16 bits are used; a keyed 8-round additive
Feistel network in 16-bit arithmetic over the mixed-radix domain ``A x B >= n`` (``A`` = power of two ~ sqrt(n),
``B = ceil(n / A)``, both >= 16) with cycle walking turns them into a bijection of ``[0, n)``.  Label shuffles are
two-level: the 16 permutations ``16g .. 16g+15`` share the 8-round bijection of their group and differ by a keyed
2-round network applied to its image (``label_permutations``); ``spatial_autocorr`` row permutations use one independent
8-round bijection each (``permutation``).  The reference (squidpy) has no such generator — it uses numpy PCG64 shuffles
(`/root/reference/src/squidpy/_utils.py:240-241`, ``gr/_nhood.py:533-538``) — so this file
does not follow a reference file; it exists so that the GPU permutation test can be checked
*bit for bit* (same permutations => same counts => same z-scores) and so that the statistical
quality of the generator can be tested against numpy's shuffles on the CPU.
"""

from __future__ import annotations

import math

import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
FEISTEL_C1 = np.uint64(0x88B5)
FEISTEL_C2 = np.uint64(0xDB2D)
N_ROUNDS = 8
MASK32 = np.uint64(0xFFFFFFFF)
MASK24 = np.uint64(0xFFFFFF)


def philox4x32_10(ctr: np.ndarray, key: tuple[int, int]) -> np.ndarray:
    """Vectorised Philox4x32-10.  ``ctr``: (..., 4) uint32; ``key``: two 32-bit ints."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c[0]
        p1 = PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


FEISTEL_GROUP = 16  # permutations 16g .. 16g+15 of a label shuffle share the group bijection pi_g


def round_keys(seed: int, perms: np.ndarray, lib: int = 0, tags: tuple[int, ...] = (0, 1)) -> np.ndarray:
    """Round keys, shape (len(perms), 4 * len(tags)) uint32, for global (permutation or group) indices ``perms``:
    Philox4x32-10 of the counters (index_lo, index_hi, lib, tag) — sqgr_rng.h: round_keys_tagged.  Tags (0, 1): the
    independent 8-round bijection of an index; (2, 3): the group bijection; (4,): the per-permutation 2-round sigma."""
    perms = np.asarray(perms, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = (seed & 0xFFFFFFFF, seed >> 32)
    out = []
    for j in tags:
        ctr = np.stack(
            [
                (perms & MASK32).astype(np.uint32),
                (perms >> np.uint64(32)).astype(np.uint32),
                np.full(perms.shape, lib, dtype=np.uint32),
                np.full(perms.shape, j, dtype=np.uint32),
            ],
            axis=-1,
        )
        out.append(philox4x32_10(ctr, key))
    return np.concatenate(out, axis=-1)


def domain_dims(n: int) -> tuple[int, int, int]:
    """Mixed-radix domain A x B >= n: A = power of two ~ sqrt(n), B = ceil(n / A), both >= 16, plus
    Bmask = 2**ceil(log2 B) - 1 (sqgr_rng.h: make_domain)."""
    r = math.isqrt(n - 1) + 1 if n > 1 else n  # ceil(sqrt(n))
    A = 16
    while A < r:
        A <<= 1
    B = max(16, -(-n // A))
    m = 1
    while m < B:
        m <<= 1
    return A, B, m - 1


MASK16 = np.uint64(0xFFFF)


def _F(v: np.ndarray, k: np.ndarray | np.uint64, bits: int) -> np.ndarray:
    """16-bit round function (sqgr_rng.h: feistel_F1 / feistel_F2), arithmetic modulo 2**16: multiply-add, xor-shift,
    multiply; the top ``bits`` bits are the result."""
    x = (v * FEISTEL_C1 + (k & MASK16)) & MASK16
    x = x ^ (x >> np.uint64(7))
    x = (x * FEISTEL_C2) & MASK16
    return x >> np.uint64(16 - bits)


def _rounds(a: np.ndarray, b: np.ndarray, A: int, B: int, Bmask: int, key) -> tuple[np.ndarray, np.ndarray]:
    """The 8 alternating additive rounds; ``key(r)`` returns the round key(s) of round r (low 16 bits are used)."""
    A1, B64 = np.uint64(A - 1), np.uint64(B)
    abits, bbits = int(A).bit_length() - 1, int(Bmask + 1).bit_length() - 1
    for r in range(0, N_ROUNDS, 2):
        a = (a + _F(b, key(r), abits)) & A1
        t = b + _F(a, key(r + 1), bbits)
        t = np.where(t >= B64, t - B64, t)
        b = np.where(t >= B64, t - B64, t)
    return a, b


def permutation_batch(n: int, rks: np.ndarray) -> np.ndarray:
    """pi[p, i] = image of i under the cycle-walked bijection of [0, n) keyed by ``rks[p]``: (P, 8) -> (P, n) int64."""
    P = rks.shape[0]
    if n <= 1:
        return np.zeros((P, n), dtype=np.int64)
    A, B, Bmask = domain_dims(n)
    B64 = np.uint64(B)
    keys = rks.astype(np.uint64) & MASK16
    key = lambda r: keys[:, r : r + 1]  # noqa: E731
    x = np.tile(np.arange(n, dtype=np.uint64), (P, 1))
    a, b = _rounds(x // B64, x % B64, A, B, Bmask, key)
    x = a * B64 + b
    bad = x >= np.uint64(n)
    while bad.any():
        a2, b2 = _rounds(a, b, A, B, Bmask, key)
        a, b = np.where(bad, a2, a), np.where(bad, b2, b)
        x = a * B64 + b
        bad = x >= np.uint64(n)
    return x.astype(np.int64)


def group_keys(seed: int, groups: np.ndarray, lib: int = 0) -> np.ndarray:
    """(len(groups), 8) keys of the group bijections pi_g (sqgr_rng.h: group_keys)."""
    return round_keys(seed, groups, lib, tags=(2, 3))


def sigma_keys(seed: int, perms: np.ndarray, lib: int = 0) -> np.ndarray:
    """(len(perms), 2) keys of the per-permutation 2-round networks sigma_p (sqgr_rng.h: sigma_keys)."""
    return round_keys(seed, perms, lib, tags=(4,))[:, :2]


def _sigma(a: np.ndarray, b: np.ndarray, A: int, B: int, Bmask: int, sk: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """sigma_p: b <- (b + F_B(a, k0)) mod B, then a <- (a + F_A(b, k1)) mod A (sqgr_rng.h: sigma_rounds)."""
    A1, B64 = np.uint64(A - 1), np.uint64(B)
    abits, bbits = int(A).bit_length() - 1, int(Bmask + 1).bit_length() - 1
    t = b + _F(a, sk[:, 0:1], bbits)
    t = np.where(t >= B64, t - B64, t)
    b = np.where(t >= B64, t - B64, t)
    a = (a + _F(b, sk[:, 1:2], abits)) & A1
    return a, b


def grouped_permutation_batch(n: int, gks: np.ndarray, sks: np.ndarray, x: np.ndarray | None = None) -> np.ndarray:
    """Two-level label permutations: pi[p, i] = sigma(sks[p]) o pi(gks[p]) (i), both cycle-walked into [0, n)
    (sqgr_rng.h: grouped_perm).  ``gks``: (P, 8) — row p holds the keys of permutation p's GROUP; ``sks``: (P, 2).
    ``x``: evaluate only these ranks (default: all of [0, n))."""
    P = gks.shape[0]
    if n <= 1:
        return np.zeros((P, n if x is None else len(x)), dtype=np.int64)
    A, B, Bmask = domain_dims(n)
    B64 = np.uint64(B)
    keys = gks.astype(np.uint64) & MASK16
    key = lambda r: keys[:, r : r + 1]  # noqa: E731
    sk = sks.astype(np.uint64) & MASK16
    x = np.tile(np.arange(n, dtype=np.uint64) if x is None else np.asarray(x, dtype=np.uint64), (P, 1))
    a, b = _rounds(x // B64, x % B64, A, B, Bmask, key)
    bad = a * B64 + b >= np.uint64(n)
    while bad.any():
        a2, b2 = _rounds(a, b, A, B, Bmask, key)
        a, b = np.where(bad, a2, a), np.where(bad, b2, b)
        bad = a * B64 + b >= np.uint64(n)
    a, b = _sigma(a, b, A, B, Bmask, sk)
    bad = a * B64 + b >= np.uint64(n)
    while bad.any():
        a2, b2 = _sigma(a, b, A, B, Bmask, sk)
        a, b = np.where(bad, a2, a), np.where(bad, b2, b)
        bad = a * B64 + b >= np.uint64(n)
    return (a * B64 + b).astype(np.int64)


def label_permutations(n: int, seed: int, perms: np.ndarray, lib: int = 0, x: np.ndarray | None = None) -> np.ndarray:
    """(len(perms), n) label-shuffle permutations of the global permutation indices ``perms`` (only the images of the
    ranks ``x`` when given)."""
    perms = np.asarray(perms, dtype=np.int64)
    return grouped_permutation_batch(n, group_keys(seed, perms // FEISTEL_GROUP, lib), sigma_keys(seed, perms, lib), x)


def independent_label_permutations(n: int, seed: int, perms: np.ndarray, lib: int = 0) -> np.ndarray:
    """(len(perms), n) label-shuffle permutations of the INDEPENDENT variant of the device generator (SQGR_SHUFFLE_INDEPENDENT=1,
    sqgr_nhood.hip: k_shuffle_indep): every permutation its own 8-round bijection keyed by ``round_keys(seed, perm, lib)`` — no
    group bijection shared by 16 permutations, no sigma network."""
    return permutation_batch(n, round_keys(seed, np.asarray(perms, dtype=np.int64), lib))


def permutation(n: int, rk: np.ndarray) -> np.ndarray:
    """pi with pi[i] = image of i under the cycle-walked bijection of [0, n); int64 (n,)."""
    return permutation_batch(n, np.asarray(rk)[None, :])[0]


def shuffled_labels(
    labels: np.ndarray, seed: int, perm: int, lib_ids: np.ndarray | None = None, n_libs: int = 0
) -> np.ndarray:
    """Label vector of global permutation ``perm``.

    The base vector is taken sorted by label (inside each library): a uniformly random arrangement of
    a multiset does not depend on the base order, and a sorted base turns the device's label lookup into a
    binary search over K boundaries instead of a memory gather:  out[i] = sort(labels)[pi(rank_i)]."""
    labels = np.asarray(labels)
    if lib_ids is None:
        return np.sort(labels)[label_permutations(len(labels), seed, np.array([perm]))[0]]
    out = np.empty_like(labels)
    for lib in range(n_libs):
        idx = np.where(lib_ids == lib)[0]
        out[idx] = np.sort(labels[idx])[label_permutations(len(idx), seed, np.array([perm]), lib=lib)[0]]
    return out


AUTOCORR_STREAM = 0x5A17  # "library" word of the Philox counter for spatial_autocorr permutations


def autocorr_permutation(n: int, seed: int, perm: int) -> np.ndarray:
    """Row permutation ``idx`` of global permutation ``perm`` for spatial_autocorr (sqgr_autocorr.hip:k_perm_indices)."""
    return permutation(n, round_keys(seed, np.array([perm]), lib=AUTOCORR_STREAM)[0])
