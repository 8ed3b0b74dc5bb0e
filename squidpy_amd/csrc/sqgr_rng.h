// Counter-based permutation generator for the on-device label shuffles (gfx950).
//
//   round keys : Philox4x32-10( counter = (perm_lo, perm_hi, library, j), key = (seed_lo, seed_hi) ),
//                j = 0,1  ->  8 x 32-bit keys per (seed, global permutation index, library)
//   bijection  : 8-round alternating additive Feistel network on the mixed-radix domain A x B >= n
//                (A = power of two ~ sqrt(n), B = ceil(n / A), both >= 16) whose round function uses only
//                full-rate 24-bit multiplies (v_mul_u32_u24) and xor-shifts, cycle-walked into [0, n).
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
constexpr uint32_t FEISTEL_C1 = 0xD2511Fu, FEISTEL_C2 = 0xCD9E8Du;
constexpr int FEISTEL_ROUNDS = 8;

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

__host__ __device__ inline void round_keys(uint64_t seed, uint64_t perm, uint32_t lib, uint32_t rk[8]) {
    for (uint32_t j = 0; j < 2; ++j) {
        uint32_t c[4] = {(uint32_t)perm, (uint32_t)(perm >> 32), lib, j};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        rk[4 * j + 0] = c[0]; rk[4 * j + 1] = c[1]; rk[4 * j + 2] = c[2]; rk[4 * j + 3] = c[3];
    }
}

// round function: two 24x24-bit multiplies (low 32 bits of each product, v_mul_u32_u24 reads only the low
// 24 bits of its operands) with an xor-shift between; 16 well-mixed bits out.
__device__ __forceinline__ uint32_t feistel_F(uint32_t v, uint32_t k) {
    uint32_t u = __umul24(v ^ k, FEISTEL_C1);
    u ^= u >> 15;
    uint32_t w = __umul24(u, FEISTEL_C2);
    return w >> 16;
}

// Mixed-radix domain A x B >= n, x <-> (a, b), x = a*B + b:  A = power of two ~ sqrt(n) (so the a-rounds reduce with
// one AND and are exactly uniform), B = ceil(n / A); both >= 16 and < 2^16.  The excess A*B - n is < A, so cycle
// walking almost never iterates (no wave divergence), unlike a power-of-two domain whose excess can approach n.
struct FeistelDomain {
    uint32_t n;  // target domain [0, n)
    uint32_t A;  // radix of the high digit
    uint32_t B;  // radix of the low digit
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
    return d;
}

// image of x (< n) under the cycle-walked keyed bijection of [0, n): 8 alternating additive Feistel rounds
//   a <- (a + F(b,k_r)) mod A  (A = 2^m: one AND) ;  b <- (b + (F(a,k_r+1) * B >> 16)) mod B
__device__ __forceinline__ uint32_t feistel_perm_ab(uint32_t a, uint32_t b, const FeistelDomain& d,
                                                    const uint32_t* __restrict__ rk, uint32_t* hi_digit = nullptr) {
    uint32_t x;
    do {
#pragma unroll
        for (int r = 0; r < FEISTEL_ROUNDS; r += 2) {
            a = (a + feistel_F(b, rk[r])) & (d.A - 1u);
            uint32_t t = b + (__umul24(feistel_F(a, rk[r + 1]), d.B) >> 16);
            b = min(t, t - d.B);  // t < 2B: subtract B when t >= B (unsigned wrap makes the other branch huge)
        }
        x = a * d.B + b;
    } while (x >= d.n);
    if (hi_digit) *hi_digit = a;  // x == a * d.B + b
    return x;
}

__device__ __forceinline__ uint32_t feistel_perm(uint32_t x, const FeistelDomain& d, const uint32_t* __restrict__ rk) {
    const uint32_t a = x / d.B;
    return feistel_perm_ab(a, x - a * d.B, d, rk);
}

}  // namespace sqgr
