// Counter-based permutation generator for the on-device label shuffles (gfx950).
//
//   round keys : Philox4x32-10( counter = (index_lo, index_hi, library, j), key = (seed_lo, seed_hi) ),
//                j = 0,1  ->  8 x 32-bit keys per (seed, global permutation index, library)      [feistel_perm]
//                j = 2,3  ->  8 keys per (seed, permutation GROUP index, library)                [group bijection pi_g]
//                j = 4    ->  2 keys per (seed, global permutation index, library)               [per-permutation sigma_p]
//   bijection  : 8-round alternating additive Feistel network on the mixed-radix domain A x B >= n
//                (A = power of two ~ sqrt(n), B = ceil(n / A), both >= 16) in 16-bit arithmetic — two permutations
//                per instruction on the packed-16 VALU — cycle-walked into [0, n); the low 16 bits of the Philox
//                words are the round keys.
//
// Label shuffles (nhood_enrichment, ligrec) use a TWO-LEVEL construction: the FEISTEL_GROUP = 16 permutations with
// global indices 16g .. 16g+15 share one strong bijection pi_g (the 8-round network, keys j = 2,3 of index g) and differ
// by a cheap 2-round network sigma_p (keys j = 4 of index p) applied to its image:  perm_p = sigma_p o pi_g, both
// cycle-walked into [0, n).  Each perm_p is as uniform as pi_g (a fixed bijection after a uniform one is uniform); the
// permutations of one group are dependent, which leaves every per-permutation statistic untouched and enters the
// permutation-test moments only through the label contingency tables of (L o sigma_p, L o sigma_p') — tested to be
// those of independent arrangements (tests/test_devrng.py, tools/null_moments.py).  One pi_g evaluation per spot serves
// 16 label bytes: 3x fewer VALU operations per label than 16 independent 8-round evaluations.
// Row permutations of spatial_autocorr keep one independent 8-round bijection per permutation (feistel_perm).
//
// oracle/devrng.py restates this file bit for bit; tests/test_devrng.py checks both the Philox
// known-answer vectors and the statistical quality (uniformity over S_n for small n, agreement of
// permutation-test moments with numpy's PCG64 shuffles).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sqgr {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;
constexpr uint32_t FEISTEL_C1 = 0x88B5u, FEISTEL_C2 = 0xDB2Du;  // odd 16-bit multipliers of the round function
#ifndef SQGR_FEISTEL_ROUNDS
#define SQGR_FEISTEL_ROUNDS 8  // overridable only for the round-count study of tools/null_moments.py
#endif
constexpr int FEISTEL_ROUNDS = SQGR_FEISTEL_ROUNDS;

__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c[0];
        uint64_t p1 = (uint64_t)PHILOX_M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
}

__host__ __device__ inline void round_keys_tagged(uint64_t seed, uint64_t index, uint32_t lib, uint32_t tag0, uint32_t rk[8]) {
    for (uint32_t j = 0; j < 2; ++j) {
        uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), lib, tag0 + j};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        rk[4 * j + 0] = c[0]; rk[4 * j + 1] = c[1]; rk[4 * j + 2] = c[2]; rk[4 * j + 3] = c[3];
    }
}
__host__ __device__ inline void round_keys(uint64_t seed, uint64_t perm, uint32_t lib, uint32_t rk[8]) {
    round_keys_tagged(seed, perm, lib, 0u, rk);
}
// two-level label shuffles: permutations 16g .. 16g+15 share the group bijection keyed by group_keys(g)
constexpr int FEISTEL_GROUP = 16;
__host__ __device__ inline void group_keys(uint64_t seed, uint64_t group, uint32_t lib, uint32_t rk[8]) {
    round_keys_tagged(seed, group, lib, 2u, rk);
}
__host__ __device__ inline void sigma_keys(uint64_t seed, uint64_t perm, uint32_t lib, uint32_t rk[2]) {
    uint32_t c[4] = {(uint32_t)perm, (uint32_t)(perm >> 32), lib, 4u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    rk[0] = c[0];
    rk[1] = c[1];
}

// 16-bit lanes: every quantity of the bijection (digits, keys, round function) is a 16-bit value, so two permutations
// are evaluated per instruction with gfx950's packed-16 VALU ops (v_pk_mul_lo_u16, v_pk_add_u16, v_pk_lshrrev_b16,
// v_pk_min_u16, plus plain 32-bit bitwise ops acting on both halves).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// round function: 16-bit multiply-add / xor-shift / multiply, of which the TOP `16 - sh` bits are returned (the best mixed
// ones, and already reduced to the digit's power-of-two range: no mask afterwards); arithmetic modulo 2^16.
// 5 packed ops: v_pk_mad_u16, v_pk_lshrrev_b16, v_xor_b32, v_pk_mul_lo_u16, v_pk_lshrrev_b16.
__device__ __forceinline__ u16x2 feistel_F2(u16x2 v, u16x2 k, u16x2 sh) {
    u16x2 x = v * (u16x2)(FEISTEL_C1) + k;
    x ^= x >> (u16x2)(7);
    x *= (u16x2)(FEISTEL_C2);
    return x >> sh;
}
__host__ __device__ inline uint32_t feistel_F1(uint32_t v, uint32_t k, uint32_t sh) {  // the same function, one permutation
    uint32_t x = (v * FEISTEL_C1 + (k & 0xFFFFu)) & 0xFFFFu;
    x ^= x >> 7;
    x = (x * FEISTEL_C2) & 0xFFFFu;
    return x >> sh;
}

// Mixed-radix domain A x B >= n, x <-> (a, b), x = a*B + b:  A = power of two ~ sqrt(n) (the a-rounds reduce with one
// AND and are exactly uniform), B = ceil(n / A), Bmask = 2^ceil(log2 B) - 1; 16 <= B <= A <= 2^14 (n <= 2^27), so every
// intermediate fits 16 bits.  The excess A*B - n is < A: cycle walking almost never iterates (no wave divergence).
struct FeistelDomain {
    uint32_t n;      // target domain [0, n)
    uint32_t A;      // radix of the high digit (power of two)
    uint32_t B;      // radix of the low digit
    uint32_t Bmask;  // smallest all-ones mask >= B - 1 ... (2^ceil(log2 B) - 1)
    uint32_t ash;    // 16 - log2(A): the round function's shift for the high digit
    uint32_t bsh;    // 16 - log2(Bmask + 1): ... for the low digit
};

__host__ __device__ inline uint32_t isqrt_ceil(uint32_t n) {
    uint32_t r = 0;
    while ((uint64_t)r * r < n) ++r;  // host-side only in practice (domain construction)
    return r;
}

__host__ __device__ inline FeistelDomain make_domain(uint32_t n) {
    FeistelDomain d;
    d.n = n;
    const uint32_t r = isqrt_ceil(n);
    uint32_t a = 16u;
    while (a < r) a <<= 1;
    d.A = a;
    uint32_t b = (n + d.A - 1) / d.A;
    d.B = b < 16u ? 16u : b;
    uint32_t m = 1u;
    while (m < d.B) m <<= 1;
    d.Bmask = m - 1u;
    d.ash = 16u;
    for (uint32_t t = d.A; t > 1u; t >>= 1) --d.ash;
    d.bsh = 16u;
    for (uint32_t t = m; t > 1u; t >>= 1) --d.bsh;
    return d;
}

// one pass of the 8 alternating additive rounds on NP packed pairs (= 2*NP permutations) in lock-step:
//   a <- (a + F_A(b, k_r)) mod A                    (A = 2^m: one AND; F_A < A)
//   b <- (b + F_B(a, k_r+1)) mod B                  (F_B <= Bmask < 2B, sum < 3B: two conditional subtractions)
// The round chain of one pair is strictly dependent and packed-16 results need a wait state before use; NP >= 2
// independent chains interleave and fill those slots.   pk[i][r]: round-r keys of pair i's two permutations, packed
// (low half: first permutation) — see k_keygen.
template <int NP>
__device__ __forceinline__ void feistel_rounds(u16x2 (&a)[NP], u16x2 (&b)[NP], const FeistelDomain& d,
                                               const uint32_t* const (&pk)[NP]) {
    const u16x2 am = (u16x2)((unsigned short)(d.A - 1u));
    const u16x2 ash = (u16x2)((unsigned short)d.ash), bsh = (u16x2)((unsigned short)d.bsh);
    const u16x2 BB = (u16x2)((unsigned short)d.B);
#pragma unroll
    for (int r = 0; r < FEISTEL_ROUNDS; r += 2) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const u16x2 k0 = __builtin_bit_cast(u16x2, pk[i][r]);
            a[i] = (a[i] + feistel_F2(b[i], k0, ash)) & am;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const u16x2 k1 = __builtin_bit_cast(u16x2, pk[i][r + 1]);
            u16x2 t = b[i] + feistel_F2(a[i], k1, bsh);
            t = __builtin_elementwise_min(t, (u16x2)(t - BB));  // unsigned wrap makes the wrong branch huge
            b[i] = __builtin_elementwise_min(t, (u16x2)(t - BB));
        }
    }
}

// sigma_p: two additive rounds, low digit first —  b <- (b + F_B(a, k_0)) mod B;  a <- (a + F_A(b, k_1)) mod A.
// (Low digit first: the label of a rank is decided by its high digit almost alone, so the shift of the high digit must
// depend on both input digits for neighbouring ranks to part.)
template <int NP>
__device__ __forceinline__ void sigma_rounds(u16x2 (&a)[NP], u16x2 (&b)[NP], const FeistelDomain& d,
                                             const uint32_t* const (&pk)[NP]) {
    const u16x2 am = (u16x2)((unsigned short)(d.A - 1u));
    const u16x2 ash = (u16x2)((unsigned short)d.ash), bsh = (u16x2)((unsigned short)d.bsh);
    const u16x2 BB = (u16x2)((unsigned short)d.B);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const u16x2 k0 = __builtin_bit_cast(u16x2, pk[i][0]);
        u16x2 t = b[i] + feistel_F2(a[i], k0, bsh);
        t = __builtin_elementwise_min(t, (u16x2)(t - BB));
        b[i] = __builtin_elementwise_min(t, (u16x2)(t - BB));
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const u16x2 k1 = __builtin_bit_cast(u16x2, pk[i][1]);
        a[i] = (a[i] + feistel_F2(b[i], k1, ash)) & am;
    }
}

// single-permutation form of the two-level label permutation (same arithmetic): image of x (< n) under
// sigma(sk) o pi(gk), gk = group_keys(seed, perm / 16, lib), sk = sigma_keys(seed, perm, lib)
__host__ __device__ inline uint32_t grouped_perm(uint32_t x, const FeistelDomain& d, const uint32_t* gk, const uint32_t* sk) {
    uint32_t a = x / d.B, b = x - a * d.B;
    do {
        for (int r = 0; r < FEISTEL_ROUNDS; r += 2) {
            a = (a + feistel_F1(b, gk[r], d.ash)) & (d.A - 1u);
            uint32_t t = b + feistel_F1(a, gk[r + 1], d.bsh);
            t = t >= d.B ? t - d.B : t;
            b = t >= d.B ? t - d.B : t;
        }
    } while (a * d.B + b >= d.n);
    do {
        uint32_t t = b + feistel_F1(a, sk[0], d.bsh);
        t = t >= d.B ? t - d.B : t;
        b = t >= d.B ? t - d.B : t;
        a = (a + feistel_F1(b, sk[1], d.ash)) & (d.A - 1u);
    } while (a * d.B + b >= d.n);
    return a * d.B + b;
}

// single-permutation form (same arithmetic): image of x (< n)
__host__ __device__ inline uint32_t feistel_perm(uint32_t x, const FeistelDomain& d, const uint32_t* rk) {
    uint32_t a = x / d.B, b = x - a * d.B;
    do {
        for (int r = 0; r < FEISTEL_ROUNDS; r += 2) {
            a = (a + feistel_F1(b, rk[r], d.ash)) & (d.A - 1u);
            uint32_t t = b + feistel_F1(a, rk[r + 1], d.bsh);
            t = t >= d.B ? t - d.B : t;
            b = t >= d.B ? t - d.B : t;
        }
        x = a * d.B + b;
    } while (x >= d.n);
    return x;
}

}  // namespace sqgr
