"""TEST INFRASTRUCTURE — not product code.

CPU (numpy) restatement of the *device* permutation generator used by the HIP path when
``rng="philox"`` (``squidpy_amd/csrc/sqgr_rng.h``): Philox4x32-10 derives eight 32-bit round
keys per (seed, permutation index, library); a keyed 8-round additive Feistel network over the
mixed-radix domain ``A x B >= n`` (``A`` = power of two ~ sqrt(n), ``B = ceil(n / A)``, both >= 16) with cycle
walking turns them into a bijection of ``[0, n)``.  The reference (squidpy) has no such generator — it uses numpy PCG64 shuffles
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
FEISTEL_C1 = np.uint64(0xD2511F)
FEISTEL_C2 = np.uint64(0xCD9E8D)
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


def round_keys(seed: int, perms: np.ndarray, lib: int = 0) -> np.ndarray:
    """Round keys, shape (len(perms), 8) uint32, for global permutation indices ``perms``."""
    perms = np.asarray(perms, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = (seed & 0xFFFFFFFF, seed >> 32)
    out = []
    for j in (0, 1):
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


def domain_dims(n: int) -> tuple[int, int]:
    """Mixed-radix domain A x B >= n: A = power of two ~ sqrt(n), B = ceil(n / A), both >= 16
    (sqgr_rng.h: make_domain)."""
    r = math.isqrt(n - 1) + 1 if n > 1 else n  # ceil(sqrt(n))
    A = 16
    while A < r:
        A <<= 1
    B = max(16, -(-n // A))
    return A, B


def _F(v: np.ndarray, k: np.uint64) -> np.ndarray:
    t = (v ^ k) & MASK24
    u = (t * FEISTEL_C1) & MASK32
    u ^= u >> np.uint64(15)
    w = ((u & MASK24) * FEISTEL_C2) & MASK32
    return w >> np.uint64(16)


def feistel(a: np.ndarray, b: np.ndarray, A: int, B: int, rk: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """One application of the keyed bijection of [0, A) x [0, B).  ``rk``: (8,) uint32."""
    A64, B64 = np.uint64(A), np.uint64(B)
    for r in range(0, N_ROUNDS, 2):
        a = (a + _F(b, np.uint64(int(rk[r])))) & (A64 - np.uint64(1))
        b = b + ((_F(a, np.uint64(int(rk[r + 1]))) * B64) >> np.uint64(16))
        b = np.where(b >= B64, b - B64, b)
    return a, b


def permutation(n: int, rk: np.ndarray) -> np.ndarray:
    """pi with pi[i] = image of i under the cycle-walked bijection of [0, n); int64 (n,)."""
    if n <= 1:
        return np.zeros(n, dtype=np.int64)
    A, B = domain_dims(n)
    x = np.arange(n, dtype=np.uint64)
    a, b = feistel(x // np.uint64(B), x % np.uint64(B), A, B, rk)
    x = a * np.uint64(B) + b
    bad = x >= np.uint64(n)
    while bad.any():
        a[bad], b[bad] = feistel(a[bad], b[bad], A, B, rk)
        x = a * np.uint64(B) + b
        bad = x >= np.uint64(n)
    return x.astype(np.int64)


def permutation_batch(n: int, rks: np.ndarray) -> np.ndarray:
    """Same as :func:`permutation` for many key sets at once: ``rks`` (P, 8) -> (P, n) int64 (vectorised over keys)."""
    P = rks.shape[0]
    if n <= 1:
        return np.zeros((P, n), dtype=np.int64)
    A, B = domain_dims(n)
    A64, B64 = np.uint64(A), np.uint64(B)
    keys = rks.astype(np.uint64)

    def apply(a: np.ndarray, b: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        for r in range(0, N_ROUNDS, 2):
            a = (a + _F(b, keys[:, r : r + 1])) & (A64 - np.uint64(1))
            b = b + ((_F(a, keys[:, r + 1 : r + 2]) * B64) >> np.uint64(16))
            b = np.where(b >= B64, b - B64, b)
        return a, b

    x = np.tile(np.arange(n, dtype=np.uint64), (P, 1))
    a, b = apply(x // B64, x % B64)
    x = a * B64 + b
    bad = x >= np.uint64(n)
    while bad.any():
        a2, b2 = apply(a, b)
        a, b = np.where(bad, a2, a), np.where(bad, b2, b)
        x = a * B64 + b
        bad = x >= np.uint64(n)
    return x.astype(np.int64)


def shuffled_labels(
    labels: np.ndarray, seed: int, perm: int, lib_ids: np.ndarray | None = None, n_libs: int = 0
) -> np.ndarray:
    """Label vector of global permutation ``perm``.

    The base vector is taken sorted by label (inside each library): a uniformly random arrangement of
    a multiset does not depend on the base order, and a sorted base turns the device's label lookup into a
    binary search over K boundaries instead of a memory gather:  out[i] = sort(labels)[pi(rank_i)]."""
    labels = np.asarray(labels)
    if lib_ids is None:
        rk = round_keys(seed, np.array([perm]))[0]
        return np.sort(labels)[permutation(len(labels), rk)]
    out = np.empty_like(labels)
    for lib in range(n_libs):
        idx = np.where(lib_ids == lib)[0]
        rk = round_keys(seed, np.array([perm]), lib=lib)[0]
        out[idx] = np.sort(labels[idx])[permutation(len(idx), rk)]
    return out


AUTOCORR_STREAM = 0x5A17  # "library" word of the Philox counter for spatial_autocorr permutations


def autocorr_permutation(n: int, seed: int, perm: int) -> np.ndarray:
    """Row permutation ``idx`` of global permutation ``perm`` for spatial_autocorr (sqgr_autocorr.hip:k_perm_indices)."""
    return permutation(n, round_keys(seed, np.array([perm]), lib=AUTOCORR_STREAM)[0])
