// libsqgr: numpy-compatible permutation streams on the device.
//
// Reference semantics: `spawn_generators(seed, n)` (_utils.py:240-241) gives every permutation its own PCG64 generator;
// the label test shuffles with `Generator.shuffle` (gr/_nhood.py:533-538, gr/_utils.py:207-212, gr/_ligrec.py:643-646), the
// autocorrelation test draws `Generator.permutation(N)` (gr/_ppatterns.py:270-271).  Both are the same reverse Fisher-Yates
// with masked rejection sampling over buffered 32-bit halves of the 64-bit outputs.
#include "sqgr_pcg.h"

#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sqgr {

// Bit-for-bit numpy on the device ("next" row f-1 of SURVEY.md §8): permutation p is shuffled by numpy's own
// algorithm — PCG64 (128-bit LCG, XSL-RR output, 32-bit halves buffered low first) driving the reverse
// Fisher-Yates of Generator.shuffle with masked rejection sampling — from the generator state that
// `np.random.default_rng(SeedSequence(seed).spawn(n)[p])` starts in (computed by numpy on the host, 32 bytes per
// permutation).  Two kernels: k_pcg_shuffle_wave (default, one WAVE per permutation, further down) and the simpler
// k_pcg_shuffle (SQGR_PCG_KERNEL=lane: ONE THREAD PER PERMUTATION, each on its own column of a [position][permutation]
// matrix; the access to row i is coalesced across threads, the access to row j is the scattered one) — kept as an
// independent implementation the tests compare against.
struct Pcg64 {
    uint64_t lo, hi, inc_lo, inc_hi;
    uint32_t buf;
    bool has;
};

__device__ __forceinline__ uint64_t pcg64_next64(Pcg64& g) {
    const uint64_t ML = 0x4385DF649FCCF645ull, MH = 0x2360ED051FC65DA4ull;  // PCG_DEFAULT_MULTIPLIER_128
    const uint64_t plo = g.lo * ML;
    const uint64_t phi = __umul64hi(g.lo, ML) + g.lo * MH + g.hi * ML;
    const uint64_t nlo = plo + g.inc_lo;
    const uint64_t nhi = phi + g.inc_hi + (nlo < plo ? 1ull : 0ull);
    g.lo = nlo;
    g.hi = nhi;
    const uint64_t x = nhi ^ nlo;
    const unsigned rot = (unsigned)(nhi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

__device__ __forceinline__ uint32_t pcg64_next32(Pcg64& g) {
    if (g.has) {
        g.has = false;
        return g.buf;
    }
    const uint64_t v = pcg64_next64(g);
    g.has = true;
    g.buf = (uint32_t)(v >> 32);
    return (uint32_t)v;
}

constexpr int PCG_UNROLL = 8;   // swaps replayed in registers per trip (2*PCG_UNROLL loads in flight)
constexpr int PCG_BLOCK = 32;   // steps whose draws are generated together (multiple of PCG_UNROLL)

// W[pos * stride + p]: column p = the array numpy would shuffle for permutation p (positions grouped by library,
// libraries in category order == the order `_shuffle_group` visits them, gr/_utils.py:207-212).
//
// Two things keep the lanes of a wave busy although every lane runs its own sequential Fisher-Yates:
//  * rejection sampling is decoupled from the steps: the draws j of PCG_BLOCK consecutive steps are produced in one
//    per-lane loop (a lane that needs fewer draws idles only until the slowest lane of the wave has its PCG_BLOCK
//    draws, instead of after every single step) and parked in LDS;
//  * the draws depend on the generator only, so the 2*PCG_UNROLL loads of a trip are issued before any swap is
//    applied and the swaps are replayed in registers.  Step t exchanges positions i-t and j_t; i-t is above
//    everything later steps touch, so its value is final after step t; a j position may recur (j_e == j_t, or
//    j_e == i-t for e < t) and then takes the value the earlier step left there instead of the stale load.
template <typename T, bool ARANGE>
__global__ __launch_bounds__(64) void k_pcg_shuffle(int64_t n, int n_libs, const uint32_t* __restrict__ lib_off,
                                                    const T* __restrict__ base_pos, const uint64_t* __restrict__ states,
                                                    int64_t P, int64_t stride, T* __restrict__ W) {
    __shared__ uint32_t s_j[PCG_BLOCK][64];
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= P) return;
    Pcg64 g;
    g.hi = states[4 * p + 0];
    g.lo = states[4 * p + 1];
    g.inc_hi = states[4 * p + 2];
    g.inc_lo = states[4 * p + 3];
    g.has = false;
    g.buf = 0;
    T* col = W + p;
    for (int64_t e = 0; e < n; ++e) col[e * stride] = ARANGE ? (T)e : base_pos[e];
    for (int l = 0; l < n_libs; ++l) {
        const uint32_t off = lib_off[l];
        const uint32_t m = lib_off[l + 1] - off;
        if (m < 2) continue;
        T* sub = col + (int64_t)off * stride;
        uint32_t mask = m - 1;  // smallest all-ones mask >= i, maintained as i decreases
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t i = m - 1;
        while (i >= (uint32_t)PCG_BLOCK) {
            // ---- draws of steps i, i-1, ..., i-PCG_BLOCK+1 (numpy: `while ((v = next_uint32() & mask) > max);`)
            {
                int t = 0;
                uint32_t it = i;
                if ((mask >> 1) >= it) mask >>= 1;
                if (g.has) {  // high half left over from the previous block
                    g.has = false;
                    const uint32_t c = g.buf & mask;
                    if (c <= it) {
                        s_j[t][lane] = c;
                        ++t;
                        --it;
                        if ((mask >> 1) >= it) mask >>= 1;
                    }
                }
                while (t < PCG_BLOCK) {
                    const uint64_t v = pcg64_next64(g);
                    const uint32_t c0 = (uint32_t)v & mask;
                    if (c0 <= it) {
                        s_j[t][lane] = c0;
                        ++t;
                        --it;
                        if ((mask >> 1) >= it) mask >>= 1;
                    }
                    if (t < PCG_BLOCK) {
                        const uint32_t c1 = (uint32_t)(v >> 32) & mask;
                        if (c1 <= it) {
                            s_j[t][lane] = c1;
                            ++t;
                            --it;
                            if ((mask >> 1) >= it) mask >>= 1;
                        }
                    } else {  // the block is complete after the low half: numpy keeps the high half buffered
                        g.has = true;
                        g.buf = (uint32_t)(v >> 32);
                    }
                }
            }
            // ---- apply: PCG_UNROLL swaps per trip
            constexpr int U = PCG_UNROLL;
#pragma unroll 1
            for (int t0 = 0; t0 < PCG_BLOCK; t0 += U) {
                const uint32_t ib = i - t0;
                uint32_t j[U];
                T vi[U], vj[U];
#pragma unroll
                for (int t = 0; t < U; ++t) j[t] = s_j[t0 + t][lane];
#pragma unroll
                for (int t = 0; t < U; ++t) {
                    vi[t] = sub[(int64_t)(ib - t) * stride];
                    vj[t] = sub[(int64_t)j[t] * stride];
                }
                T at_i[U], at_j[U];  // values left at position ib-t resp. j_t by step t
#pragma unroll
                for (int t = 0; t < U; ++t) {
                    T cur_i = vi[t], cur_j = vj[t];
#pragma unroll
                    for (int e = 0; e < t; ++e) {  // a later e overrides an earlier one
                        cur_i = (j[e] == ib - t) ? at_j[e] : cur_i;
                        cur_j = (j[e] == j[t]) ? at_j[e] : cur_j;
                    }
                    at_i[t] = cur_j;
                    at_j[t] = cur_i;
                }
                // j stores in step order (a recurring position keeps the last one), then the final i stores on top
#pragma unroll
                for (int t = 0; t < U; ++t) sub[(int64_t)j[t] * stride] = at_j[t];
#pragma unroll
                for (int t = 0; t < U; ++t) sub[(int64_t)(ib - t) * stride] = at_i[t];
            }
            i -= PCG_BLOCK;
            // the mask was advanced to step i-PCG_BLOCK's bound already (idempotent update at the top of the next block)
        }
        for (; i >= 1; --i) {
            if ((mask >> 1) >= i) mask >>= 1;
            uint32_t j;
            do {
                j = pcg64_next32(g) & mask;
            } while (j > i);
            const T a = sub[(int64_t)i * stride], b = sub[(int64_t)j * stride];
            sub[(int64_t)i * stride] = b;
            sub[(int64_t)j * stride] = a;
        }
    }
}

// ---- one WAVE per permutation ------------------------------------------------------------------------------------
// The sequential chain of Fisher-Yates is only apparent:
//  (1) the raw generator outputs are a pure function of the stream position (LCG jump-ahead: state_k = A_k*state +
//      G_k*inc with A_k = M^k, G_k = 1 + M + ... + M^(k-1) mod 2^128), so the 64 lanes produce the next 64 raw 32-bit
//      draws at once;
//  (2) whether draw d is accepted depends on earlier draws only through the NUMBER of earlier acceptances (< 64): a
//      candidate <= i-64 is accepted and one > i rejected regardless; the few in between are decided in order;
//  (3) the accepted swaps of a trip touch disjoint positions unless two draws coincide or a draw hits the i side of a
//      step of the same trip.  Then (probability ~ 64*64/i per trip) the steps that share a position are replayed in
//      order on a register image of the touched positions (lane q: values at i-q and at j_q), all others stay plain
//      exchanges.
// Both paths consume the identical draw sequence as numpy (32-bit halves, low first; a library that ends mid-chunk leaves
// the remaining draws to the next one; a mask change or the last 192 steps take the draws one by one).
// Layout: R[p][row_stride] (a permutation's array is contiguous).  ~45 steps per trip; measured 12x the
// one-thread-per-permutation kernel at Squidpy's default n_perms = 1000 (1e5 spots: 4.0 ms vs 48 ms).
struct U128 {
    uint64_t hi, lo;
};
__device__ __forceinline__ U128 mul128_lo(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
    return r;
}
__device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

constexpr int PCGW_TAB = 34;      // jump distances 0..33 (64 halves starting at an odd half reach output 33)
constexpr int PCGW_HASH = 2048;   // duplicate filter of the fast path
constexpr uint32_t PCGW_MIN_SPAN = 192;  // positions below i that must be in the LDS window before a trip

// lane's raw 32-bit draw number `lane` after stream position (s, half); sk = the LCG state it was taken from
__device__ __forceinline__ uint32_t pcgw_draw(const U128& s, uint32_t half, int lane, const uint64_t (*sA)[2],
                                              const uint64_t (*sD)[2], U128& sk) {
    const uint32_t hl = half + (uint32_t)lane;
    const int kk = 1 + (int)(hl >> 1);
    U128 a, d;
    a.hi = sA[kk][0];
    a.lo = sA[kk][1];
    d.hi = sD[kk][0];
    d.lo = sD[kk][1];
    sk = add128(mul128_lo(a, s), d);
    const uint64_t x = sk.hi ^ sk.lo;
    const unsigned rot = (unsigned)(sk.hi >> 58);
    const uint64_t out = (x >> rot) | (x << ((64u - rot) & 63u));
    return (hl & 1u) ? (uint32_t)(out >> 32) : (uint32_t)out;
}

// consume `used` of the 64 draws: (s, half) move on; sk = the per-lane states pcgw_draw returned for the old position
__device__ __forceinline__ void pcgw_advance(U128& s, uint32_t& half, const U128& sk, uint32_t used) {
    const uint32_t hu = half + used;
    const uint32_t dq = hu >> 1;
    if (dq > 0u) {
        const int src = (int)(2u * dq - 1u - half);
        s.hi = readlane64(sk.hi, src);
        s.lo = readlane64(sk.lo, src);
    }
    half = hu & 1u;
}

template <typename T, bool ARANGE>
__global__ __launch_bounds__(64) void k_pcg_shuffle_wave(int64_t n, int64_t row_stride, int n_libs,
                                                         const uint32_t* __restrict__ lib_off, const T* __restrict__ base_pos,
                                                         const uint64_t* __restrict__ states, const uint64_t* __restrict__ jump,
                                                         int64_t P, T* __restrict__ R, int force_slow, uint32_t WS) {
    // Window: the top WS positions [wlo, i] of the array being shuffled live in LDS (ring, index pos & (WS-1)); positions
    // are finalised (written to R) when they are the i side of a step, everything else they see stays in LDS.  An
    // array of <= WS elements is resident as a whole.  This takes the read-after-write on the line just stored off the
    // critical path of every trip.
    extern __shared__ unsigned char s_window[];
    T* win = reinterpret_cast<T*>(s_window);
    const uint32_t WM = WS - 1u;
    __shared__ uint64_t sA[PCGW_TAB][2], sD[PCGW_TAB][2];
    __shared__ uint32_t sJ[64];
    __shared__ uint32_t sH[PCGW_HASH];
    const int lane = threadIdx.x;
    for (int h = lane; h < PCGW_HASH; h += 64) sH[h] = 0;
    uint32_t epoch = 0;
    // a block walks permutations blockIdx.x, blockIdx.x + gridDim.x, ... (the grid may be capped so that the rows being
    // shuffled at any one time stay cache resident)
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
    U128 s, inc;
    s.hi = states[4 * p + 0];
    s.lo = states[4 * p + 1];
    inc.hi = states[4 * p + 2];
    inc.lo = states[4 * p + 3];
    if (lane < PCGW_TAB) {
        U128 a, g;
        a.hi = jump[4 * lane + 0];
        a.lo = jump[4 * lane + 1];
        g.hi = jump[4 * lane + 2];
        g.lo = jump[4 * lane + 3];
        const U128 d = mul128_lo(g, inc);
        sA[lane][0] = a.hi;
        sA[lane][1] = a.lo;
        sD[lane][0] = d.hi;
        sD[lane][1] = d.lo;
    }
    T* row = R + p * row_stride;
    for (int64_t e = lane; e < n; e += 64) row[e] = ARANGE ? (T)e : base_pos[e];
    uint32_t half = 0;   // 1: the next draw is the high half of the 64-bit output after `s`
    U128 sk;             // per lane: generator state behind this lane's draw
    uint32_t raw = pcgw_draw(s, half, lane, sA, sD, sk);  // the next 64 raw 32-bit draws, one per lane
    for (int l = 0; l < n_libs; ++l) {
        const uint32_t off = lib_off[l];
        const uint32_t m = lib_off[l + 1] - off;
        if (m < 2) continue;
        T* sub = row + off;
        uint32_t mask = m - 1;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t i = m - 1;
        uint32_t wlo = m;  // window = positions [wlo, i], empty at first
        while (i >= 1) {
            // Lanes of this wave exchange data through LDS and through the row in global memory (a position stored by one
            // lane in trip k is loaded by another lane in trip k+1).  A wavefront executes its memory instructions in
            // order, so wavefront-scope ordering costs no instruction; the fence states the requirement and keeps the
            // compiler from carrying values across trips.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (wlo > 0u && i + 1u - wlo < PCGW_MIN_SPAN) {  // refill to capacity: positions [lo2, wlo)
                const uint32_t lo2 = (i + 1u > WS) ? i + 1u - WS : 0u;
                for (uint32_t q = lo2 + (uint32_t)lane; q < wlo; q += 64u) win[q & WM] = sub[q];
                wlo = lo2;
            }
            uint32_t used = 64, nacc = 0, it = i;
            const bool general = force_slow || i < 192u || (mask >> 1) >= i - 64u;
            if (!general) {
                // the bound stays above i-64 and the mask cannot change: a candidate <= i-64 is accepted, one > i rejected
                // whatever happened before it; the few in between are decided in order from the running count
                const uint32_t c = raw & mask;
                const bool sure = c <= i - 64u;
                const uint64_t ambm = __ballot(!sure && c <= i);
                uint64_t accm = __ballot(sure);
                for (uint64_t rem = ambm; rem != 0ull; rem &= rem - 1ull) {
                    const int dd = __builtin_ctzll(rem);
                    const uint32_t cd = (uint32_t)__builtin_amdgcn_readlane((int)c, dd);
                    const uint32_t before = (uint32_t)__popcll(accm & ((1ull << dd) - 1ull));
                    if (cd <= i - before) accm |= 1ull << dd;
                }
                const bool acc = (accm >> lane) & 1ull;
                const uint32_t t = __builtin_amdgcn_mbcnt_hi((uint32_t)(accm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)accm, 0u));
                nacc = (uint32_t)__popcll(accm);
                it = i - nacc;
                bool conflict = (ambm & accm) != 0ull;  // such a j may be the i side of a step of this very trip
                if (!conflict) {
                    // coinciding draws: every accepted lane tags its bucket; a lane that reads back another lane's tag shares
                    // the bucket with it.  With a single such lane the only possible twin is the bucket's winner.
                    ++epoch;
                    const uint32_t tag = (epoch << 6) | (uint32_t)lane;
                    const uint32_t h = (c * 2654435761u) >> 21;
                    if (acc) sH[h] = tag;
                    asm volatile("" ::: "memory");  // the read-back must see the other lanes' stores, not be forwarded
                    const uint32_t got = acc ? sH[h] : tag;
                    const uint64_t cm = __ballot(got != tag);
                    if (cm != 0ull) {
                        bool dup = false;
                        if ((cm & (cm - 1ull)) == 0ull) {
                            const uint32_t cw = (uint32_t)__shfl((int)c, (int)(got & 63u));
                            dup = got != tag && cw == c;
                        } else {
                            for (int e = 0; e < 64; ++e) {
                                const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)c, e);
                                if (((accm >> e) & 1ull) && acc && e < lane && ce == c) dup = true;
                            }
                        }
                        conflict = __ballot(dup) != 0ull;
                    }
                }
                if (!conflict) {
                    // ---- all swaps of the trip are independent.  The next chunk's draws are computed while the loads fly.
                    T va = 0, vb = 0;
                    if (acc) {
                        va = win[(i - t) & WM];
                        vb = (c >= wlo) ? win[c & WM] : sub[c];
                    }
                    pcgw_advance(s, half, sk, 64u);
                    U128 sk2;
                    const uint32_t raw2 = pcgw_draw(s, half, lane, sA, sD, sk2);
                    if (acc) {
                        sub[i - t] = vb;  // final
                        if (c >= wlo) win[c & WM] = va; else sub[c] = va;
                    }
                    i = it;
                    raw = raw2;
                    sk = sk2;
                    continue;
                }
                if (acc) sJ[t] = c;  // step-indexed list for the replay
            } else {
                // ---- sequential acceptance over the 64 raw draws (wave-uniform scalar code): mask changes, the last
                // steps of a library, tiny arrays
                uint32_t t = 0;
                for (int dd = 0; dd < 64; ++dd) {
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)raw, dd) & mask;
                    if (c <= it) {
                        sJ[t] = c;
                        ++t;
                        --it;
                        if ((mask >> 1) >= it) mask >>= 1;
                        if (it == 0u) {
                            used = (uint32_t)dd + 1u;
                            break;
                        }
                    }
                }
                nacc = t;
            }
            // ---- exact replay of steps 0..nacc-1 (step q: positions i-q and sJ[q]) on a register image: lane q holds the
            // value at i-q (vI) and at its j (vJ)
            if (nacc > 0u) {
                asm volatile("" ::: "memory");
                const bool live = (uint32_t)lane < nacc;
                const uint32_t j = live ? sJ[lane] : 0xFFFFFFFFu;
                uint32_t vI = 0, vJ = 0;
                if (live) {
                    vI = (uint32_t)win[(i - (uint32_t)lane) & WM];
                    vJ = (uint32_t)((j >= wlo) ? win[j & WM] : sub[j]);
                }
                // slot of position j_q: 64 + q' if it is the i side of step q', else the first step with that j
                uint32_t canon = (uint32_t)lane;
                const uint32_t ilow = i - (nacc - 1u);
                const bool in_i = live && j >= ilow;
                if (in_i) canon = 64u + (i - j);
                for (uint32_t e = 0; e + 1u < nacc; ++e) {
                    const uint32_t je = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)e);
                    if (live && !in_i && e < (uint32_t)lane && je == j && canon == (uint32_t)lane) canon = e;
                }
                // steps that share a slot with another step run in order; all others are plain exchanges
                const uint64_t refm = __ballot(live && canon != (uint32_t)lane);
                uint64_t inv = refm;
                for (uint64_t rem = refm; rem != 0ull; rem &= rem - 1ull) {
                    const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)canon, __builtin_ctzll(rem));
                    inv |= 1ull << (slot >= 64u ? slot - 64u : slot);
                }
                if (live && !((inv >> lane) & 1ull)) {
                    const uint32_t tmp = vI;
                    vI = vJ;
                    vJ = tmp;
                }
                for (uint64_t rem = inv; rem != 0ull; rem &= rem - 1ull) {
                    const int q = __builtin_ctzll(rem);
                    const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)canon, q);
                    const uint32_t va = (uint32_t)__builtin_amdgcn_readlane((int)vI, q);
                    if (slot >= 64u) {
                        const int q2 = (int)(slot - 64u);
                        const uint32_t vb = (uint32_t)__builtin_amdgcn_readlane((int)vI, q2);
                        vI = (lane == q) ? vb : vI;
                        vI = (lane == q2) ? va : vI;
                    } else {
                        const uint32_t vb = (uint32_t)__builtin_amdgcn_readlane((int)vJ, (int)slot);
                        vI = (lane == q) ? vb : vI;
                        vJ = (lane == (int)slot) ? va : vJ;
                    }
                }
                if (live) {
                    sub[i - (uint32_t)lane] = (T)vI;  // final
                    if (canon == (uint32_t)lane) {
                        if (j >= wlo) win[j & WM] = (T)vJ; else sub[j] = (T)vJ;
                    }
                }
            }
            i = it;
            pcgw_advance(s, half, sk, used);
            raw = pcgw_draw(s, half, lane, sA, sD, sk);
        }
        if (lane == 0) sub[0] = win[0];  // position 0 is never the i side of a step: its last value is still in LDS
    }
    }
}

// ---- the bucketed replay: coalesced traffic instead of one random HBM sector per swap side --------------------------
// At 1e6 positions the wave kernel above moves two random 64-byte sectors per swap (152 MB per permutation for a 1 MB row).
// The swaps of numpy's Fisher-Yates can be replayed in another order without changing the result (argument and numpy-level
// proof of the order: oracle/pcg_bucket.py, tests/test_pcg_bucket_cpu.py):
//  * phase f = the steps whose i lies in [f*S, (f+1)*S) (S = 2^logS positions, <= 65536), range r = j / S;
//  * position i is read only by step i itself and by earlier steps e > i with j_e == i — those lie in range r == f; no later
//    step touches i again (j_e <= e < i).  So a phase may apply its swaps RANGE BY RANGE: first the window range r == f (both
//    sides inside the phase's own S positions), then every range r < f in any order, time order inside a range.
// k_pcg_draws_bucketed (G, one wave per permutation) generates the draws with the exact acceptance logic of the wave kernel and
// appends (j mod S | (i mod S) << 16) to time-ordered lists per (phase, range), in 64-record blocks: records
// [perm][phase][block][64], directory entry per block = range | count << 8 | ordinal-within-range << 16.
// k_pcg_apply_bucketed (A, one 1024-lane workgroup per permutation) walks the phases from the top: window and one range at a
// time in LDS (coalesced 64 KB loads and stores), the records of a (phase, range) list applied 1024 at a time in ROUNDS: a
// record goes when it is the earliest pending record of the chunk on its j slot (hashed tags, atomic max of epoch | 1023 - lane)
// and — in the window range — no earlier pending record writes to its i slot; the others wait for the next round (~3.5 rounds
// per chunk at 1e6 positions).  Traffic per permutation: ranges 16 phases x ~0.5 MB x 2 + records 2 x 4.3 MB.
constexpr int PCGB_MAX_RANGES = 64;   // ranges per library (lane r of the generator wave keeps range r's cursors)
constexpr int PCGB_RING = 128;        // staging ring per range (records)
constexpr int PCGB_SLOTS = 3328;      // conflict tags of the apply kernel at 65536-position windows (two buffers: rounds alternate); shorter
                                      // windows leave more of the 160 KB of LDS to the tags: PcgBucketGeom::slots
constexpr int PCGB_CHUNK_BLOCKS = 32; // 64-record blocks per chunk of the apply kernel: two records per lane
constexpr int PCGB_THREADS = 1024;

struct PcgBucketGeom {
    int logS;              // log2 of the phase / range length
    int n_ranges;          // max over libraries of ceil(m / S)
    int bcap;              // blocks per phase: S / 64 + n_ranges (rounded up when S < 64)
    int phases;            // phases of all libraries of one permutation
    int slots;             // conflict tags per buffer of the apply kernel (two buffers)
    int qcap;              // k_pcg_apply_claims: queue entries in use (<= PCGQ_CAP; SQGR_PCG_QUEUE_CAP: tests of the spill path)
};

__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }

__global__ __launch_bounds__(64) void k_pcg_draws_bucketed(int n_libs, const uint32_t* __restrict__ lib_off, const uint32_t* __restrict__ lib_phase,
                                                           const uint64_t* __restrict__ states, const uint64_t* __restrict__ jump, int64_t P,
                                                           PcgBucketGeom geo, uint32_t* __restrict__ recs, uint32_t* __restrict__ dir,
                                                           uint32_t* __restrict__ nblk, int force_slow) {
    extern __shared__ uint32_t s_stage[];  // [n_ranges][PCGB_RING]
    __shared__ uint64_t sA[PCGW_TAB][2], sD[PCGW_TAB][2];
    __shared__ uint32_t sJ[64];
    __shared__ unsigned long long s_lanes[PCGB_MAX_RANGES];  // per range: the lanes whose record goes there (zero between appends)
    const int lane = threadIdx.x;
    s_lanes[lane] = 0ull;
    const uint32_t logS = (uint32_t)geo.logS, SM = (1u << logS) - 1u;
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    U128 s, inc;
    s.hi = states[4 * p + 0];
    s.lo = states[4 * p + 1];
    inc.hi = states[4 * p + 2];
    inc.lo = states[4 * p + 3];
    if (lane < PCGW_TAB) {
        U128 a, g;
        a.hi = jump[4 * lane + 0];
        a.lo = jump[4 * lane + 1];
        g.hi = jump[4 * lane + 2];
        g.lo = jump[4 * lane + 3];
        const U128 d = mul128_lo(g, inc);
        sA[lane][0] = a.hi;
        sA[lane][1] = a.lo;
        sD[lane][0] = d.hi;
        sD[lane][1] = d.lo;
    }
    uint32_t* const rec_p = recs + (size_t)p * geo.phases * geo.bcap * 64;
    uint32_t* const dir_p = dir + (size_t)p * geo.phases * geo.bcap;
    uint32_t* const nblk_p = nblk + (size_t)p * geo.phases;
    // lane r keeps the cursors of range r: records staged, ring head, blocks flushed in this phase
    uint32_t c_cnt = 0, c_head = 0, c_ord = 0;
    uint32_t nb = 0;   // blocks written in the current phase (uniform)
    uint32_t ph = 0;   // global index of the current phase (uniform)
    auto flush_block = [&](uint32_t r0, uint32_t head, uint32_t count) {  // uniform arguments
        const uint32_t v = s_stage[r0 * PCGB_RING + ((head + (uint32_t)lane) & (PCGB_RING - 1))];
        uint32_t* dst = rec_p + ((size_t)ph * geo.bcap + nb) * 64;
        const uint32_t ordinal = lane_get(c_ord, r0);
        if ((uint32_t)lane < count) dst[lane] = v;
        if (lane == 0) dir_p[(size_t)ph * geo.bcap + nb] = r0 | (count << 8) | (ordinal << 16);
        if ((uint32_t)lane == r0) ++c_ord;
        ++nb;
    };
    auto end_phase = [&](uint32_t f) {  // the partial blocks of ranges 0..f, then the phase's block count
        for (uint32_t r0 = 0; r0 <= f; ++r0) {
            const uint32_t cnt = lane_get(c_cnt, r0);
            if (cnt > 0u) flush_block(r0, lane_get(c_head, r0), cnt);
        }
        if (lane == 0) nblk_p[ph] = nb;
        c_cnt = 0;
        c_head = 0;
        c_ord = 0;
        nb = 0;
    };
    // append the records of the lanes with `in` set (lane order == time order) to their ranges' rings; full blocks go out.
    // The lanes of a range find each other through a 64-bit lane mask per range in LDS (one ds_or per record, one read): the
    // rank of a record among its range's records of this trip is a count of the mask bits below its lane, the owner lane of a
    // range adds the population count to its cursor — no loop over the ranges (a first version walked them one by one and made
    // this kernel issue-bound at ~2000 clk per trip).
    auto append = [&](bool in, uint32_t iloc, uint32_t jj) {
        const uint32_t r = jj >> logS;
        const uint32_t rec = (jj & SM) | (iloc << 16);
        if (in) atomicOr(&s_lanes[r], 1ull << lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned long long mk = in ? s_lanes[r] : 0ull;
        const unsigned long long own = (lane < geo.n_ranges) ? s_lanes[lane] : 0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < geo.n_ranges) s_lanes[lane] = 0ull;  // re-armed for the next call
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        const uint32_t cur = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r << 2), (int)(c_cnt | (c_head << 8)));  // the cursors of MY range
        if (in) s_stage[r * PCGB_RING + (((cur >> 8) + (cur & 0xffu) + rank) & (PCGB_RING - 1))] = rec;
        c_cnt += (uint32_t)__popcll(own);
        uint64_t full = __ballot(c_cnt >= 64u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the rings are read back by other lanes of this wave
        for (; full != 0ull; full &= full - 1ull) {
            const uint32_t r0 = (uint32_t)__builtin_ctzll(full);
            flush_block(r0, lane_get(c_head, r0), 64u);
            if ((uint32_t)lane == r0) {
                c_cnt -= 64u;
                c_head = (c_head + 64u) & (PCGB_RING - 1);
            }
        }
    };
    uint32_t half = 0;
    U128 sk;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t raw = pcgw_draw(s, half, lane, sA, sD, sk);
    for (int l = 0; l < n_libs; ++l) {
        const uint32_t off = lib_off[l];
        const uint32_t m = lib_off[l + 1] - off;
        if (m < 2) continue;
        uint32_t mask = m - 1;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t i = m - 1;
        uint32_t f_cur = i >> logS;
        ph = lib_phase[l] + f_cur;
        while (i >= 1) {
            uint32_t used = 64, nacc = 0, it = i;
            bool valid = false;
            uint32_t ipos = 0, jj = 0;
            const bool general = force_slow || i < 192u || (mask >> 1) >= i - 64u;
            if (!general) {  // (see k_pcg_shuffle_wave: sure / ambiguous candidates, decided from the running count)
                const uint32_t c = raw & mask;
                const bool sure = c <= i - 64u;
                const uint64_t ambm = __ballot(!sure && c <= i);
                uint64_t accm = __ballot(sure);
                for (uint64_t rem = ambm; rem != 0ull; rem &= rem - 1ull) {
                    const int dd = __builtin_ctzll(rem);
                    const uint32_t cd = lane_get(c, (uint32_t)dd);
                    const uint32_t before = (uint32_t)__popcll(accm & ((1ull << dd) - 1ull));
                    if (cd <= i - before) accm |= 1ull << dd;
                }
                valid = (accm >> lane) & 1ull;
                const uint32_t t = __builtin_amdgcn_mbcnt_hi((uint32_t)(accm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)accm, 0u));
                nacc = (uint32_t)__popcll(accm);
                it = i - nacc;
                ipos = i - t;
                jj = c;
            } else {
                uint32_t t = 0;
                for (int dd = 0; dd < 64; ++dd) {
                    const uint32_t c = lane_get(raw, (uint32_t)dd) & mask;
                    if (c <= it) {
                        if (lane == 0) sJ[t] = c;
                        ++t;
                        --it;
                        if ((mask >> 1) >= it) mask >>= 1;
                        if (it == 0u) {
                            used = (uint32_t)dd + 1u;
                            break;
                        }
                    }
                }
                nacc = t;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                valid = (uint32_t)lane < nacc;
                ipos = i - (uint32_t)lane;
                jj = valid ? sJ[lane] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (nacc > 0u) {
                const uint32_t f_last = (it + 1u) >> logS;  // phase of the trip's last step
                for (uint32_t f = f_cur;; --f) {
                    append(valid && (ipos >> logS) == f, ipos & SM, jj);
                    if (f == f_last) break;
                    end_phase(f);
                    --ph;
                }
                f_cur = f_last;
            }
            i = it;
            if (i >= 1u && (i >> logS) != f_cur) {  // the next step opens a new phase
                end_phase(f_cur);
                --ph;
                f_cur = i >> logS;
            }
            pcgw_advance(s, half, sk, used);
            raw = pcgw_draw(s, half, lane, sA, sD, sk);
        }
        end_phase(f_cur);  // f_cur == 0 here: the library's last phase
    }
}

// ---- the draw generator, 128 raw draws per trip (round 6) ---------------------------------------------------------------
// k_pcg_draws_bucketed spends most of a trip on bookkeeping that does not depend on the number of draws (uniform branches,
// cursor updates, the stream position: ~250 instructions of which 12 are the 128-bit multiply) and computes every LCG state in
// two lanes, one per 32-bit half.  Here lane l advances to state l + 1 once and takes BOTH halves — slot 2l (low half) and
// slot 2l + 1 (high half), numpy's order — so a trip consumes 128 draws (127 when the low half of the first state was used
// before: after a full trip the stream always stands at a state boundary).  The acceptance, the running count, the phase
// crossings and the time order of the records appended to a range's list are the ones of the 64-draw kernel, on pairs of lane
// masks: slot (l, h) comes before (l', h') when l < l' or l == l' and h < h'.
constexpr int PCGW_TAB2 = 65;     // jump distances 0..64
// One wavefront per permutation is bound by the latencies of its own instruction stream, so the kernel lives on occupancy (measured:
// 8 KB more LDS per wavefront cost the 64-draw kernel 49.7 -> 73.4 ms per 8192 permutations).  LDS per wavefront here: a 64-record
// ring per range (a block goes out the moment it is complete, so a ring never holds more than one) + two lane masks per range +
// the step list of the one-by-one path = 4.9 KB at 16 ranges (64-draw kernel: 10 KB); the jump constants of lane l never change
// (state l + 1 of every trip) and stay in registers.

__device__ __forceinline__ uint64_t pcgw_draw2(const U128& s, const U128& a, const U128& d, U128& sk) {
    sk = add128(mul128_lo(a, s), d);
    const uint64_t x = sk.hi ^ sk.lo;
    const unsigned rot = (unsigned)(sk.hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
// consume `used` slots starting at slot `half`: lane dq - 1 holds state dq
__device__ __forceinline__ void pcgw_advance2(U128& s, uint32_t& half, const U128& sk, uint32_t used) {
    const uint32_t hu = half + used;
    const uint32_t dq = hu >> 1;
    if (dq > 0u) {
        s.hi = readlane64(sk.hi, (int)dq - 1);
        s.lo = readlane64(sk.lo, (int)dq - 1);
    }
    half = hu & 1u;
}
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {  // bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
static size_t pcg_draws2_lds_bytes(int n_ranges) { return (size_t)n_ranges * (64 * 4 + 2 * 8); }

__global__ __launch_bounds__(64) void k_pcg_draws_bucketed2(int n_libs, const uint32_t* __restrict__ lib_off, const uint32_t* __restrict__ lib_phase,
                                                            const uint64_t* __restrict__ states, const uint64_t* __restrict__ jump, int64_t P,
                                                            PcgBucketGeom geo, uint32_t* __restrict__ recs, uint32_t* __restrict__ dir,
                                                            uint32_t* __restrict__ nblk, int force_slow) {
    extern __shared__ uint32_t s_stage[];  // [n_ranges][64], then the lane masks
    unsigned long long* const s_lanes0 = reinterpret_cast<unsigned long long*>(s_stage + (size_t)geo.n_ranges * 64);  // per range: lanes whose slot-0 ...
    unsigned long long* const s_lanes1 = s_lanes0 + geo.n_ranges;                                                     // ... | slot-1 record goes there
    __shared__ uint32_t sJ[128];
    const int lane = threadIdx.x;
    if (lane < geo.n_ranges) {
        s_lanes0[lane] = 0ull;
        s_lanes1[lane] = 0ull;
    }
    const uint32_t logS = (uint32_t)geo.logS, SM = (1u << logS) - 1u;
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    U128 s, inc;
    s.hi = states[4 * p + 0];
    s.lo = states[4 * p + 1];
    inc.hi = states[4 * p + 2];
    inc.lo = states[4 * p + 3];
    U128 ja, jd;  // state_(l+1) = ja * s + jd: A_(l+1) and G_(l+1) * inc
    {
        U128 g;
        ja.hi = jump[4 * (lane + 1) + 0];
        ja.lo = jump[4 * (lane + 1) + 1];
        g.hi = jump[4 * (lane + 1) + 2];
        g.lo = jump[4 * (lane + 1) + 3];
        jd = mul128_lo(g, inc);
    }
    uint32_t* const rec_p = recs + (size_t)p * geo.phases * geo.bcap * 64;
    uint32_t* const dir_p = dir + (size_t)p * geo.phases * geo.bcap;
    uint32_t* const nblk_p = nblk + (size_t)p * geo.phases;
    uint32_t c_cnt = 0, c_ord = 0;  // lane r: range r's staged records (< 64 between appends), blocks flushed in this phase
    uint32_t nb = 0;   // blocks written in the current phase (uniform)
    uint32_t ph = 0;   // global index of the current phase (uniform)
    auto flush_block = [&](uint32_t r0, uint32_t count) {  // uniform arguments
        const uint32_t v = s_stage[r0 * 64u + (uint32_t)lane];
        uint32_t* dst = rec_p + ((size_t)ph * geo.bcap + nb) * 64;
        const uint32_t ordinal = lane_get(c_ord, r0);
        if ((uint32_t)lane < count) dst[lane] = v;
        if (lane == 0) dir_p[(size_t)ph * geo.bcap + nb] = r0 | (count << 8) | (ordinal << 16);
        if ((uint32_t)lane == r0) ++c_ord;
        ++nb;
    };
    auto end_phase = [&](uint32_t f) {  // the partial blocks of ranges 0..f, then the phase's block count
        for (uint32_t r0 = 0; r0 <= f; ++r0) {
            const uint32_t cnt = lane_get(c_cnt, r0);
            if (cnt > 0u) flush_block(r0, cnt);
        }
        if (lane == 0) nblk_p[ph] = nb;
        c_cnt = 0;
        c_ord = 0;
        nb = 0;
    };
    // append the slot-0 / slot-1 records of the lanes with in0 / in1 set to their ranges' rings in slot order.  A range may receive up
    // to 128 records on top of the < 64 it holds: they go in passes of one block — write the records that fall into the block being
    // filled, send the complete blocks out, go on with the next 64 positions.
    auto append = [&](bool in0, uint32_t iloc0, uint32_t jj0, bool in1, uint32_t iloc1, uint32_t jj1) {
        const uint32_t r0 = jj0 >> logS, r1 = jj1 >> logS;
        const unsigned long long bit = 1ull << lane;
        if (in0) atomicOr(&s_lanes0[r0], bit);
        if (in1) atomicOr(&s_lanes1[r1], bit);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unsigned long long a0 = 0ull, a1 = 0ull, b0 = 0ull, b1 = 0ull;
        if (in0) {
            a0 = s_lanes0[r0];
            a1 = s_lanes1[r0];
        }
        if (in1) {
            b0 = s_lanes0[r1];
            b1 = s_lanes1[r1];
        }
        unsigned long long own0 = 0ull, own1 = 0ull;
        if (lane < geo.n_ranges) {
            own0 = s_lanes0[lane];
            own1 = s_lanes1[lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < geo.n_ranges) {  // re-armed for the next call
            s_lanes0[lane] = 0ull;
            s_lanes1[lane] = 0ull;
        }
        // position in the range's list counted from the start of the block being filled: records staged + slots before this one
        const uint32_t pos0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r0 << 2), (int)c_cnt) + mbcnt64(a0) + mbcnt64(a1);
        const uint32_t pos1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r1 << 2), (int)c_cnt) + mbcnt64(b0) + mbcnt64(b1) + (uint32_t)((b0 >> lane) & 1ull);
        const uint32_t rec0 = (jj0 & SM) | (iloc0 << 16), rec1 = (jj1 & SM) | (iloc1 << 16);
        uint32_t tot = c_cnt + (uint32_t)__popcll(own0) + (uint32_t)__popcll(own1);  // lane r: range r's records, staged + new
        for (uint32_t base = 0;; base += 64u) {
            if (in0 && pos0 - base < 64u) s_stage[r0 * 64u + (pos0 - base)] = rec0;
            if (in1 && pos1 - base < 64u) s_stage[r1 * 64u + (pos1 - base)] = rec1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the rings are read back by other lanes of this wave
            uint64_t full = __ballot(tot >= base + 64u);
            if (full == 0ull) break;
            for (; full != 0ull; full &= full - 1ull) flush_block((uint32_t)__builtin_ctzll(full), 64u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        c_cnt = tot & 63u;
    };
    uint32_t half = 0;
    U128 sk;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint64_t raw = pcgw_draw2(s, ja, jd, sk);
    for (int l = 0; l < n_libs; ++l) {
        const uint32_t off = lib_off[l];
        const uint32_t m = lib_off[l + 1] - off;
        if (m < 2) continue;
        uint32_t mask = m - 1;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t i = m - 1;
        uint32_t f_cur = i >> logS;
        ph = lib_phase[l] + f_cur;
        while (i >= 1) {
            uint32_t used = 128u - half, nacc = 0, it = i;
            bool valid0 = false, valid1 = false;
            uint32_t ipos0 = 0, ipos1 = 0, jj0 = 0, jj1 = 0;
            const bool general = force_slow || i < 320u || (mask >> 1) >= i - 128u;
            if (!general) {
                // the bound stays above i - 128 and the mask cannot change: a candidate <= i - 128 is accepted, one > i rejected
                // whatever happened before it; the few in between are decided in slot order from the running count
                const uint32_t c0 = (uint32_t)raw & mask, c1 = (uint32_t)(raw >> 32) & mask;
                const bool live0 = !(lane == 0 && half != 0u);  // slot 0 of the first state was consumed before
                const bool sure0 = live0 && c0 <= i - 128u, sure1 = c1 <= i - 128u;
                uint64_t amb0 = __ballot(live0 && !sure0 && c0 <= i), amb1 = __ballot(!sure1 && c1 <= i);
                uint64_t acc0 = __ballot(sure0), acc1 = __ballot(sure1);
                while ((amb0 | amb1) != 0ull) {
                    const uint32_t l0 = amb0 ? (uint32_t)__builtin_ctzll(amb0) : 64u, l1 = amb1 ? (uint32_t)__builtin_ctzll(amb1) : 64u;
                    if (l0 <= l1) {  // slot (l0, 0) comes before slot (l1, 1)
                        const uint64_t below = (1ull << l0) - 1ull;
                        const uint32_t before = (uint32_t)__popcll(acc0 & below) + (uint32_t)__popcll(acc1 & below);
                        if (lane_get(c0, l0) <= i - before) acc0 |= 1ull << l0;
                        amb0 &= amb0 - 1ull;
                    } else {
                        const uint64_t below = (1ull << l1) - 1ull;
                        const uint32_t before = (uint32_t)__popcll(acc0 & (below | (1ull << l1))) + (uint32_t)__popcll(acc1 & below);
                        if (lane_get(c1, l1) <= i - before) acc1 |= 1ull << l1;
                        amb1 &= amb1 - 1ull;
                    }
                }
                valid0 = (acc0 >> lane) & 1ull;
                valid1 = (acc1 >> lane) & 1ull;
                const uint32_t t0 = mbcnt64(acc0) + mbcnt64(acc1), t1 = t0 + (valid0 ? 1u : 0u);
                nacc = (uint32_t)__popcll(acc0) + (uint32_t)__popcll(acc1);
                it = i - nacc;
                ipos0 = i - t0;
                ipos1 = i - t1;
                jj0 = c0;
                jj1 = c1;
            } else {
                uint32_t t = 0;
                const uint32_t rlo = (uint32_t)raw, rhi = (uint32_t)(raw >> 32);
                for (uint32_t dd = half; dd < 128u; ++dd) {  // slot dd = 2 * lane + (high half ? 1 : 0)
                    const uint32_t c = ((dd & 1u) ? lane_get(rhi, dd >> 1) : lane_get(rlo, dd >> 1)) & mask;
                    if (c <= it) {
                        if (lane == 0) sJ[t] = c;
                        ++t;
                        --it;
                        if ((mask >> 1) >= it) mask >>= 1;
                        if (it == 0u) {
                            used = dd + 1u - half;
                            break;
                        }
                    }
                }
                nacc = t;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                valid0 = 2u * (uint32_t)lane < nacc;       // step 2 * lane ...
                valid1 = 2u * (uint32_t)lane + 1u < nacc;  // ... and step 2 * lane + 1 of the trip
                ipos0 = i - 2u * (uint32_t)lane;
                ipos1 = ipos0 - 1u;
                jj0 = valid0 ? sJ[2 * lane] : 0u;
                jj1 = valid1 ? sJ[2 * lane + 1] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (nacc > 0u) {
                const uint32_t f_last = (it + 1u) >> logS;  // phase of the trip's last step
                for (uint32_t f = f_cur;; --f) {
                    append(valid0 && (ipos0 >> logS) == f, ipos0 & SM, jj0, valid1 && (ipos1 >> logS) == f, ipos1 & SM, jj1);
                    if (f == f_last) break;
                    end_phase(f);
                    --ph;
                }
                f_cur = f_last;
            }
            i = it;
            if (i >= 1u && (i >> logS) != f_cur) {  // the next step opens a new phase
                end_phase(f_cur);
                --ph;
                f_cur = i >> logS;
            }
            pcgw_advance2(s, half, sk, used);
            raw = pcgw_draw2(s, ja, jd, sk);
        }
        end_phase(f_cur);  // f_cur == 0 here: the library's last phase
    }
}

__device__ __forceinline__ void pcgb_copy_in(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t len, int tid) {
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0u) {
        const uint32_t n16 = len >> 4;
        for (uint32_t k = (uint32_t)tid; k < n16; k += PCGB_THREADS) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
        for (uint32_t k = (n16 << 4) + (uint32_t)tid; k < len; k += PCGB_THREADS) dst[k] = src[k];
    } else {
        for (uint32_t k = (uint32_t)tid; k < len; k += PCGB_THREADS) dst[k] = src[k];
    }
}
__device__ __forceinline__ void pcgb_copy_out(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t len, int tid) {
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0u) {
        const uint32_t n16 = len >> 4;
        for (uint32_t k = (uint32_t)tid; k < n16; k += PCGB_THREADS) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
        for (uint32_t k = (n16 << 4) + (uint32_t)tid; k < len; k += PCGB_THREADS) dst[k] = src[k];
    } else {
        for (uint32_t k = (uint32_t)tid; k < len; k += PCGB_THREADS) dst[k] = src[k];
    }
}

__global__ __launch_bounds__(PCGB_THREADS) void k_pcg_apply_bucketed(int64_t row_stride, int n_libs, const uint32_t* __restrict__ lib_off,
                                                                     const uint32_t* __restrict__ lib_phase, const uint8_t* __restrict__ base_pos,
                                                                     int64_t P, PcgBucketGeom geo, const uint32_t* __restrict__ recs,
                                                                     const uint32_t* __restrict__ dir, const uint32_t* __restrict__ nblk,
                                                                     uint8_t* __restrict__ R) {
    extern __shared__ unsigned char s_dyn[];
    const uint32_t logS = (uint32_t)geo.logS, S = 1u << logS;
    uint8_t* const Xw = s_dyn;                                            // the phase's window: positions [f*S, (f+1)*S)
    uint8_t* const Xr = s_dyn + S;                                        // one range r < f
    uint32_t* const tags = reinterpret_cast<uint32_t*>(s_dyn + 2 * (size_t)S);  // [2][PCGB_SLOTS]: rounds alternate between the buffers
    const uint32_t SLOTS = (uint32_t)geo.slots;
    uint32_t* const blist = tags + 2 * SLOTS;                             // [bcap] block | count << 16, sorted by (range, ordinal)
    __shared__ uint32_t s_wave_any[2][PCGB_THREADS / 64];                 // per round parity and wave: a record is still pending
    __shared__ uint32_t s_hist[PCGB_MAX_RANGES], s_start[PCGB_MAX_RANGES + 1];
    __shared__ uint8_t s_dirty[PCGB_MAX_RANGES];                          // range already written to the row (else: still base_pos)
    const int tid = threadIdx.x;
    const uint32_t L = (uint32_t)tid;
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const uint32_t* const rec_p = recs + (size_t)p * geo.phases * geo.bcap * 64;
        const uint32_t* const dir_p = dir + (size_t)p * geo.phases * geo.bcap;
        const uint32_t* const nblk_p = nblk + (size_t)p * geo.phases;
        uint8_t* const row = R + p * row_stride;
        for (uint32_t k = L; k < 2 * SLOTS; k += PCGB_THREADS) tags[k] = 0u;
        uint32_t epoch = 0;
        for (int l = 0; l < n_libs; ++l) {
            const uint32_t off = lib_off[l];
            const uint32_t m = lib_off[l + 1] - off;
            if (m == 0u) continue;
            if (m == 1u) {
                if (tid == 0) row[off] = base_pos[off];
                continue;
            }
            const uint32_t F = (m + S - 1u) >> logS;
            __syncthreads();
            if (tid < PCGB_MAX_RANGES) s_dirty[tid] = 0;
            for (uint32_t ff = F; ff-- > 0u;) {
                const uint32_t f = ff;
                const uint32_t ph = lib_phase[l] + f;
                const uint32_t wlen = min(S, m - (f << logS));
                __syncthreads();
                // the conflict tags hold (epoch << 11 | index): epoch restarts with every phase — a phase has at most S / 2048 chunks of
                // <= 2048 rounds each, far below 2^21, so the epoch field cannot wrap whatever the array length
                for (uint32_t k = L; k < 2 * SLOTS; k += PCGB_THREADS) tags[k] = 0u;
                epoch = 0;
                pcgb_copy_in(Xw, (s_dirty[f] ? row : base_pos) + off + ((size_t)f << logS), wlen, tid);
                if (tid < PCGB_MAX_RANGES) s_hist[tid] = 0u;
                __syncthreads();
                const uint32_t nb = nblk_p[ph];
                for (uint32_t t = L; t < nb; t += PCGB_THREADS) atomicAdd(&s_hist[dir_p[(size_t)ph * geo.bcap + t] & 0xffu], 1u);
                __syncthreads();
                if (tid == 0) {
                    uint32_t acc = 0;
                    for (uint32_t r = 0; r <= f; ++r) {
                        s_start[r] = acc;
                        acc += s_hist[r];
                    }
                    s_start[f + 1] = acc;
                }
                __syncthreads();
                for (uint32_t t = L; t < nb; t += PCGB_THREADS) {
                    const uint32_t e = dir_p[(size_t)ph * geo.bcap + t];
                    blist[s_start[e & 0xffu] + (e >> 16)] = t | (((e >> 8) & 0xffu) << 16);
                }
                __syncthreads();
                // Ranges in processing order: rr = 0 is the window range f, rr >= 1 is range rr - 1.  Two software pipelines hide
                // the global latencies a single resident workgroup cannot overlap otherwise: the records of the NEXT chunk and
                // the bytes of the NEXT range are loaded into registers while the current chunk runs its rounds.
                auto range_of = [&](uint32_t rr) { return rr == 0u ? f : rr - 1u; };
                auto next_range = [&](uint32_t rr) {  // first rr' > rr with records (f + 1: none)
                    for (++rr; rr <= f && s_hist[range_of(rr)] == 0u; ++rr) {}
                    return rr;
                };
                auto load_recs = [&](uint32_t rr, uint32_t blk0, bool& have) -> uint32_t {  // this wave's block blk0 + wave of range rr
                    have = false;
                    if (rr > f) return 0u;
                    const uint32_t r = range_of(rr), bi = blk0 + (L >> 6);
                    if (bi >= s_hist[r]) return 0u;
                    const uint32_t e = blist[s_start[r] + bi];
                    if ((L & 63u) >= (e >> 16)) return 0u;
                    have = true;
                    return rec_p[((size_t)ph * geo.bcap + (e & 0xffffu)) * 64 + (L & 63u)];
                };
                uint4 pre0, pre1, pre2, pre3;  // the next range's bytes (16-byte pieces tid, tid + 1024, ...): S <= 65536 = 4 x 1024 x 16
                pre0 = pre1 = pre2 = pre3 = make_uint4(0u, 0u, 0u, 0u);
                auto range_src = [&](uint32_t rr) { return (s_dirty[range_of(rr)] ? row : base_pos) + off + ((size_t)range_of(rr) << logS); };
                const uint32_t n16 = S >> 4;
                // only whole, 16-byte aligned ranges are prefetched (else: plain copy at the switch)
#define SQGR_PCGB_PREFETCH(RR, OK)                                                                            \
    do {                                                                                                      \
        OK = false;                                                                                           \
        const uint32_t rr__ = (RR);                                                                           \
        if (rr__ <= f && rr__ != 0u && S >= 16u) {                                                            \
            const uint4* src__ = reinterpret_cast<const uint4*>(range_src(rr__));                             \
            if ((reinterpret_cast<uintptr_t>(src__) & 15u) == 0u) {                                           \
                if ((uint32_t)tid < n16) pre0 = src__[tid];                                                   \
                if ((uint32_t)tid + 1024u < n16) pre1 = src__[tid + 1024];                                    \
                if ((uint32_t)tid + 2048u < n16) pre2 = src__[tid + 2048];                                    \
                if ((uint32_t)tid + 3072u < n16) pre3 = src__[tid + 3072];                                    \
                OK = true;                                                                                    \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
                uint32_t rr = s_hist[f] != 0u ? 0u : next_range(0u), c0 = 0;
                // a chunk = 32 blocks in time order: lane L of the workgroup holds record L (blocks c0 .. c0+15) and record 1024 + L
                // (blocks c0+16 .. c0+31)
                bool have_a = false, have_b = false;
                uint32_t rec_a = load_recs(rr, c0, have_a), rec_b = load_recs(rr, c0 + 16u, have_b);
                bool pre_ok = false;
                while (rr <= f) {
                    const uint32_t r = range_of(rr), nbr = s_hist[r];
                    const bool internal = rr == 0u;
                    uint8_t* const X2 = internal ? Xw : Xr;
                    // the chunk after this one: next chunk of the range, or the first chunk of the next range with records
                    const bool last_chunk = c0 + PCGB_CHUNK_BLOCKS >= nbr;
                    const uint32_t rr_n = last_chunk ? next_range(rr) : rr, c0_n = last_chunk ? 0u : c0 + PCGB_CHUNK_BLOCKS;
                    if (c0 == 0u) {  // entering range r
                        if (!internal) {
                            if (pre_ok) {
                                uint4* dst = reinterpret_cast<uint4*>(Xr);
                                if ((uint32_t)tid < n16) dst[tid] = pre0;
                                if ((uint32_t)tid + 1024u < n16) dst[tid + 1024] = pre1;
                                if ((uint32_t)tid + 2048u < n16) dst[tid + 2048] = pre2;
                                if ((uint32_t)tid + 3072u < n16) dst[tid + 3072] = pre3;
                            } else {
                                pcgb_copy_in(Xr, range_src(rr), S, tid);
                            }
                            __syncthreads();
                        }
                        SQGR_PCGB_PREFETCH(next_range(rr), pre_ok);  // in flight while this range is replayed
                    }
                    bool have_na = false, have_nb = false;
                    const uint32_t rec_na = load_recs(rr_n, c0_n, have_na), rec_nb = load_recs(rr_n, c0_n + 16u, have_nb);
                    bool pend_a = have_a, pend_b = have_b;
                    const uint32_t ja = rec_a & 0xffffu, ia = rec_a >> 16, jb = rec_b & 0xffffu, ib = rec_b >> 16;
                    const uint32_t hja = __umulhi(ja * 2654435761u, SLOTS), hia = __umulhi(ia * 2654435761u, SLOTS);
                    const uint32_t hjb = __umulhi(jb * 2654435761u, SLOTS), hib = __umulhi(ib * 2654435761u, SLOTS);
                    // ONE barrier per round: the tags of consecutive rounds live in different buffers (a fast wave's claims of round
                    // k + 1 cannot disturb a slow wave still reading round k; the barrier of round k + 1 protects the buffer's
                    // re-use in round k + 2), and so do the per-wave "still pending" words.  Tag = epoch | 2047 - index in the chunk.
                    // (Measured alternatives, all slower on MI355X: compacting the records the first round leaves onto the first
                    // waves — one more barrier and an LDS queue per chunk, 91 ms instead of 75 ms per 8192 permutations of 1e6
                    // positions; a branch-free round with per-lane dummy slots instead of masked lanes — 95 instructions per round
                    // instead of ~160, but every lane then issues every LDS operation of every round: 94 ms.)
                    for (;;) {
                        ++epoch;
                        uint32_t* const T = tags + (epoch & 1u) * SLOTS;
                        const uint32_t mine_a = (epoch << 11) | (2047u - L), mine_b = (epoch << 11) | (1023u - L);
                        if (pend_a) atomicMax(&T[hja], mine_a);
                        if (pend_b) atomicMax(&T[hjb], mine_b);
                        const bool wave_pending = __ballot(pend_a || pend_b) != 0ull;
                        if ((L & 63u) == 0u) s_wave_any[epoch & 1u][L >> 6] = wave_pending ? 1u : 0u;
                        __syncthreads();
                        const uint32_t wa = s_wave_any[epoch & 1u][L & (PCGB_THREADS / 64 - 1)];
                        // (every LDS read of a round is issued only by the lanes, and in the ranges, that need it: -2.7 %)
                        uint32_t ta = 0, tb = 0, tia = 0, tib = 0;
                        uint8_t va_i = 0, va_j = 0;
                        if (pend_a) {
                            ta = T[hja];
                            va_i = Xw[ia];
                            va_j = X2[ja];
                        }
                        if (pend_b) tb = T[hjb];
                        if (internal) {
                            if (pend_a) tia = T[hia];
                            if (pend_b) tib = T[hib];
                        }
                        if (__ballot(wa != 0u) == 0ull) break;  // uniform over the workgroup: every wave reads all 16 words
                        bool go_a = pend_a && ta == mine_a, go_b = pend_b && tb == mine_b;
                        // window range: no earlier pending record of the chunk may write to this record's i position
                        if (internal) {
                            go_a = go_a && ((tia >> 11) != epoch || (2047u - (tia & 2047u)) >= L);
                            go_b = go_b && ((tib >> 11) != epoch || (2047u - (tib & 2047u)) >= 1024u + L);
                        }
                        if (go_a) {
                            Xw[ia] = va_j;
                            X2[ja] = va_i;
                            pend_a = false;
                        }
                        if (go_b) {  // after this lane's own record a (had a been blocked on a shared position, so would b be)
                            const uint8_t vb_i = Xw[ib], vb_j = X2[jb];
                            Xw[ib] = vb_j;
                            X2[jb] = vb_i;
                            pend_b = false;
                        }
                    }
                    if (last_chunk && !internal) {
                        __syncthreads();
                        pcgb_copy_out(row + off + ((size_t)r << logS), Xr, S, tid);
                        if (tid == 0) s_dirty[r] = 1;
                        __syncthreads();
                    }
                    rr = rr_n;
                    c0 = c0_n;
                    rec_a = rec_na;
                    rec_b = rec_nb;
                    have_a = have_na;
                    have_b = have_nb;
                }
#undef SQGR_PCGB_PREFETCH
                pcgb_copy_out(row + off + ((size_t)f << logS), Xw, wlen, tid);  // positions of this phase are final
            }
        }
        __syncthreads();
    }
}

// ---- the replay with exact claims (round 6): bitmaps instead of hashed tags, contested records deferred ------------------
// k_pcg_apply_bucketed's rounds are spent on FALSE conflicts: 2048 records claim 3328 hashed tag slots, 46 % lose the first round
// to a record with another position, ~4.4 barrier rounds per chunk — while only 3-6 % of a chunk's records really share a
// position with another record of the chunk.  This kernel claims positions EXACTLY, one bit per position of the window (8 KB):
//   claim   every record ORs the bit(s) of its position(s) into the claim bitmap A (ds_or_rtn); a record that finds a bit already
//           set knows it is contested and sets the bit(s) in the poison bitmap B — which also tells the first arriver;
//   apply   (behind a barrier) a record with none of its positions in B swaps at once — it is the only record of the chunk on
//           them and every earlier record on them has gone; the others are DEFERRED: appended to a queue in LDS with their
//           time index (chunk << 11 | slot), their positions stay poisoned, so later records that touch them queue up behind
//           them (a queued record and a later unqueued one never share a position: they commute);
//   drain   when the list ends (the range's bytes leave LDS) or the queue is half full: the queued records go in the priority
//           rounds of the old kernel (hashed tags, atomic max of epoch | earliest time index, the i-side rule of the window
//           range) — a few hundred records on 2 x 1024 slots (they alias the claim bitmap, which is all zero between chunks);
//           each clears its words of B.
// A chunk costs two barrier intervals instead of ~4.4 rounds, a list one drain of ~4-6 rounds.  A record that finds the queue
// full stays with its lane and joins the drain that follows at once (slots a / b of the drain: robust for any stream).
constexpr int PCGQ_CAP = 1024;     // queue entries (4 + 2 bytes each)
constexpr int PCGQ_SLOTS = 1024;   // conflict tags per buffer of a drain (two buffers = the 8 KB of the claim bitmap at S = 65536)
constexpr int PCGQ_TAG_BYTES = 2 * PCGQ_SLOTS * 4;
#ifndef SQGR_PCGQ_NSUB
#define SQGR_PCGQ_NSUB 2
#endif
constexpr int PCGQ_NSUB = SQGR_PCGQ_NSUB;  // chunks per super-chunk of record loads

__host__ __device__ inline uint32_t pcgq_bitmap_bytes(uint32_t S) { return S / 8u < 16u ? 16u : S / 8u; }
__host__ __device__ inline uint32_t pcgq_claim_bytes(uint32_t S) { return pcgq_bitmap_bytes(S) < (uint32_t)PCGQ_TAG_BYTES ? (uint32_t)PCGQ_TAG_BYTES : pcgq_bitmap_bytes(S); }
static size_t pcgq_lds_bytes(uint32_t S, int bcap) {
    return 2 * (size_t)S + pcgq_claim_bytes(S) + pcgq_bitmap_bytes(S) + (size_t)PCGQ_CAP * 6 + (size_t)bcap * 4;
}

#ifdef SQGR_PCG_PROFILE
// developer build (tools/pcg_profile.sh): shader-clock time per section, counts of chunks / drains / drain rounds / queued records
__device__ unsigned long long g_pcgq_prof[16];
#define PCGQ_T0() unsigned long long t0__ = __builtin_amdgcn_s_memtime()
#define PCGQ_ACC(SLOT)                                                          \
    do {                                                                        \
        const unsigned long long t1__ = __builtin_amdgcn_s_memtime();           \
        if (tid == 0) atomicAdd(&g_pcgq_prof[SLOT], t1__ - t0__);               \
        t0__ = t1__;                                                            \
    } while (0)
#define PCGQ_CNT(SLOT, V)                                                       \
    do {                                                                        \
        if (tid == 0) atomicAdd(&g_pcgq_prof[SLOT], (unsigned long long)(V));   \
    } while (0)
#else
#define PCGQ_T0() do {} while (0)
#define PCGQ_ACC(SLOT) do {} while (0)
#define PCGQ_CNT(SLOT, V) do {} while (0)
#endif

__global__ __launch_bounds__(PCGB_THREADS) void k_pcg_apply_claims(int64_t row_stride, int n_libs, const uint32_t* __restrict__ lib_off,
                                                                   const uint32_t* __restrict__ lib_phase, const uint8_t* __restrict__ base_pos,
                                                                   int64_t P, PcgBucketGeom geo, const uint32_t* __restrict__ recs,
                                                                   const uint32_t* __restrict__ dir, const uint32_t* __restrict__ nblk,
                                                                   uint8_t* __restrict__ R) {
    extern __shared__ unsigned char s_dyn[];
    const uint32_t logS = (uint32_t)geo.logS, S = 1u << logS;
    uint8_t* const Xw = s_dyn;                                                   // the phase's window: positions [f*S, (f+1)*S)
    uint8_t* const Xr = s_dyn + S;                                               // one range r < f
    uint32_t* const A = reinterpret_cast<uint32_t*>(s_dyn + 2 * (size_t)S);      // claim bitmap of the chunk; a drain's tags [2][PCGQ_SLOTS]
    uint32_t* const Bm = A + pcgq_claim_bytes(S) / 4;                            // poison bitmap: positions of the queued records
    uint32_t* const Qrec = Bm + pcgq_bitmap_bytes(S) / 4;                        // queue: records ...
    uint16_t* const Qprio = reinterpret_cast<uint16_t*>(Qrec + PCGQ_CAP);        // ... and their time index (chunk << 11 | slot)
    uint32_t* const blist = reinterpret_cast<uint32_t*>(Qprio + PCGQ_CAP);       // [bcap] block | count << 16, sorted by (range, ordinal)
    __shared__ uint32_t s_wave_any[2][PCGB_THREADS / 64];
    __shared__ uint32_t s_hist[PCGB_MAX_RANGES], s_start[PCGB_MAX_RANGES + 1];
    __shared__ uint8_t s_dirty[PCGB_MAX_RANGES];
    __shared__ uint32_t s_qcount, s_spill;
    const int tid = threadIdx.x;
    const uint32_t L = (uint32_t)tid;
    const uint32_t claim_words = pcgq_claim_bytes(S) / 4, bm_words = pcgq_bitmap_bytes(S) / 4;
    const uint32_t qcap = (uint32_t)geo.qcap;  // a drain starts when the queue is half full; a record that finds it full stays with its lane
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const uint32_t* const rec_p = recs + (size_t)p * geo.phases * geo.bcap * 64;
        const uint32_t* const dir_p = dir + (size_t)p * geo.phases * geo.bcap;
        const uint32_t* const nblk_p = nblk + (size_t)p * geo.phases;
        uint8_t* const row = R + p * row_stride;
        __syncthreads();
        for (uint32_t k = L; k < claim_words + bm_words; k += PCGB_THREADS) A[k] = 0u;  // (Bm follows A)
        if (tid == 0) {
            s_qcount = 0u;
            s_spill = 0u;
        }
        for (int l = 0; l < n_libs; ++l) {
            const uint32_t off = lib_off[l];
            const uint32_t m = lib_off[l + 1] - off;
            if (m == 0u) continue;
            if (m == 1u) {
                if (tid == 0) row[off] = base_pos[off];
                continue;
            }
            const uint32_t F = (m + S - 1u) >> logS;
            __syncthreads();
            if (tid < PCGB_MAX_RANGES) s_dirty[tid] = 0;
            for (uint32_t ff = F; ff-- > 0u;) {
                const uint32_t f = ff;
                const uint32_t ph = lib_phase[l] + f;
                const uint32_t wlen = min(S, m - (f << logS));
                __syncthreads();
                PCGQ_T0();
                pcgb_copy_in(Xw, (s_dirty[f] ? row : base_pos) + off + ((size_t)f << logS), wlen, tid);
                if (tid < PCGB_MAX_RANGES) s_hist[tid] = 0u;
                __syncthreads();
                const uint32_t nb = nblk_p[ph];
                for (uint32_t t = L; t < nb; t += PCGB_THREADS) atomicAdd(&s_hist[dir_p[(size_t)ph * geo.bcap + t] & 0xffu], 1u);
                __syncthreads();
                if (tid == 0) {
                    uint32_t acc = 0;
                    for (uint32_t r = 0; r <= f; ++r) {
                        s_start[r] = acc;
                        acc += s_hist[r];
                    }
                    s_start[f + 1] = acc;
                }
                __syncthreads();
                for (uint32_t t = L; t < nb; t += PCGB_THREADS) {
                    const uint32_t e = dir_p[(size_t)ph * geo.bcap + t];
                    blist[s_start[e & 0xffu] + (e >> 16)] = t | (((e >> 8) & 0xffu) << 16);
                }
                __syncthreads();
                // (range order, the two software pipelines — next chunk's records, next range's bytes — as in k_pcg_apply_bucketed)
                auto range_of = [&](uint32_t rr) { return rr == 0u ? f : rr - 1u; };
                auto next_range = [&](uint32_t rr) {
                    for (++rr; rr <= f && s_hist[range_of(rr)] == 0u; ++rr) {}
                    return rr;
                };
                uint4 pre0, pre1, pre2, pre3;
                pre0 = pre1 = pre2 = pre3 = make_uint4(0u, 0u, 0u, 0u);
                auto range_src = [&](uint32_t rr) { return (s_dirty[range_of(rr)] ? row : base_pos) + off + ((size_t)range_of(rr) << logS); };
                const uint32_t n16 = S >> 4;
#define SQGR_PCGQ_PREFETCH(RR, OK)                                                                            \
    do {                                                                                                      \
        OK = false;                                                                                           \
        const uint32_t rr__ = (RR);                                                                           \
        if (rr__ <= f && rr__ != 0u && S >= 16u) {                                                            \
            const uint4* src__ = reinterpret_cast<const uint4*>(range_src(rr__));                             \
            if ((reinterpret_cast<uintptr_t>(src__) & 15u) == 0u) {                                           \
                if ((uint32_t)tid < n16) pre0 = src__[tid];                                                   \
                if ((uint32_t)tid + 1024u < n16) pre1 = src__[tid + 1024];                                    \
                if ((uint32_t)tid + 2048u < n16) pre2 = src__[tid + 2048];                                    \
                if ((uint32_t)tid + 3072u < n16) pre3 = src__[tid + 3072];                                    \
                OK = true;                                                                                    \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
                // The records of a list are loaded a SUPER-CHUNK (PCGQ_NSUB chunks) at a time, the next super-chunk's loads issued
                // right behind the wait for this one's: a chunk is two short barrier intervals, far less than an HBM round trip, and
                // the compiler's wait for conditionally issued loads is vmcnt(0) — with one chunk in flight per wait (the shape of
                // k_pcg_apply_bucketed) every chunk paid the whole latency.
                uint32_t rr = s_hist[f] != 0u ? 0u : next_range(0u), c0 = 0;
                auto load_super = [&](uint32_t rrx, uint32_t s0, uint32_t (&out)[2 * PCGQ_NSUB], uint32_t& hv) {
                    hv = 0u;
                    const bool any = rrx <= f;
                    const uint32_t r = any ? range_of(rrx) : 0u;
                    const uint32_t nbr = any ? s_hist[r] : 0u, st0 = s_start[r];
#pragma unroll
                    for (int q = 0; q < 2 * PCGQ_NSUB; ++q) {  // sub-chunk q / 2: record a (blocks s0 + 32 (q/2) + wave) | b (+ 16)
                        const uint32_t bi = s0 + 16u * (uint32_t)q + (L >> 6);
                        const bool in = bi < nbr;
                        const uint32_t e = blist[in ? st0 + bi : 0u];
                        const bool ok = in && (L & 63u) < (e >> 16);
                        const uint32_t* src = rec_p + ((size_t)ph * geo.bcap + (e & 0xffffu)) * 64 + (L & 63u);
                        const uint32_t v = *(ok ? src : rec_p);  // unconditional load: the eight of them issue back to back
                        out[q] = ok ? v : 0u;
                        hv |= ok ? (1u << q) : 0u;
                    }
                };
                uint32_t cur[2 * PCGQ_NSUB], nxt[2 * PCGQ_NSUB], cur_hv = 0u, nxt_hv = 0u;
                load_super(rr, c0, cur, cur_hv);
                PCGQ_ACC(0);
                bool pre_ok = false;
                uint32_t seq = 0;  // chunks since the last drain (5 bits of the time index)
                while (rr <= f) {
                    const uint32_t r = range_of(rr), nbr = s_hist[r];
                    const bool internal = rr == 0u;
                    uint8_t* const X2 = internal ? Xw : Xr;
                    const bool last_super = c0 + PCGQ_NSUB * PCGB_CHUNK_BLOCKS >= nbr;
                    const uint32_t rr_n = last_super ? next_range(rr) : rr, c0_n = last_super ? 0u : c0 + PCGQ_NSUB * PCGB_CHUNK_BLOCKS;
                    if (c0 == 0u) {  // entering range r
#if defined(SQGR_PCG_ABLATE) && (SQGR_PCG_ABLATE & 2)
                        if (false) {
#else
                        if (!internal) {
#endif
                            if (pre_ok) {
                                uint4* dst = reinterpret_cast<uint4*>(Xr);
                                if ((uint32_t)tid < n16) dst[tid] = pre0;
                                if ((uint32_t)tid + 1024u < n16) dst[tid + 1024] = pre1;
                                if ((uint32_t)tid + 2048u < n16) dst[tid + 2048] = pre2;
                                if ((uint32_t)tid + 3072u < n16) dst[tid + 3072] = pre3;
                            } else {
                                pcgb_copy_in(Xr, range_src(rr), S, tid);
                            }
                        }
#if !(defined(SQGR_PCG_ABLATE) && (SQGR_PCG_ABLATE & 2))
                        SQGR_PCGQ_PREFETCH(next_range(rr), pre_ok);  // in flight while this range is replayed
#endif
                        PCGQ_ACC(1);
                    }
                    // touch this super-chunk's records (the wait lands here), then put the next one's loads in flight
                    {
                        uint32_t acc = cur_hv;
#pragma unroll
                        for (int q = 0; q < 2 * PCGQ_NSUB; ++q) acc |= cur[q];
                        asm volatile("" ::"v"(acc));
                    }
                    PCGQ_ACC(12);
                    load_super(rr_n, c0_n, nxt, nxt_hv);
                    PCGQ_ACC(13);
                    // (INTERNAL_C: the window range — both sides of a record in Xw — as a compile-time constant: the other ranges, four
                    // records in five, run without its branches)
                    auto chunk = [&](auto INTERNAL_C, const uint32_t rec_a, const uint32_t rec_b, const bool have_a, const bool have_b, const bool last_chunk) {
                        constexpr bool internal = decltype(INTERNAL_C)::value;
                        const uint32_t ja = rec_a & 0xffffu, ia = rec_a >> 16, jb = rec_b & 0xffffu, ib = rec_b >> 16;
                        // a swap of a position with itself changes nothing and claims nothing
#if defined(SQGR_PCG_ABLATE) && (SQGR_PCG_ABLATE & 4)  // timing experiment (wrong results): no record claims or swaps anything
                        const bool act_a = have_a && ia == 0x12345u, act_b = have_b && ib == 0x12345u;
#else
                        const bool act_a = have_a && !(internal && ia == ja), act_b = have_b && !(internal && ib == jb);
#endif
                        const uint32_t wja = ja >> 5, bja = 1u << (ja & 31u), wia = ia >> 5, bia = 1u << (ia & 31u);
                        const uint32_t wjb = jb >> 5, bjb = 1u << (jb & 31u), wib = ib >> 5, bib = 1u << (ib & 31u);
                        // ---- claim
                        bool conf_a = false, conf_b = false;
                        {
                            uint32_t oja = 0, oia = 0, ojb = 0, oib = 0;
                            if (act_a) oja = atomicOr(&A[wja], bja);
                            if (act_b) ojb = atomicOr(&A[wjb], bjb);
                            if (internal) {
                                if (act_a) oia = atomicOr(&A[wia], bia);
                                if (act_b) oib = atomicOr(&A[wib], bib);
                            }
                            conf_a = ((oja & bja) | (oia & bia)) != 0u;
                            conf_b = ((ojb & bjb) | (oib & bib)) != 0u;
                            if (conf_a) {
                                atomicOr(&Bm[wja], bja);
                                if (internal) atomicOr(&Bm[wia], bia);
                            }
                            if (conf_b) {
                                atomicOr(&Bm[wjb], bjb);
                                if (internal) atomicOr(&Bm[wib], bib);
                            }
                        }
                        __syncthreads();  // (also: the range's bytes stored above are in place)
                        PCGQ_ACC(2);
                        PCGQ_CNT(8, 1);
                        // ---- apply or defer
                        bool def_a = false, def_b = false;
                        // (measured and dropped: the claims, poison look-ups and clears issued by every lane without branches —
                        // 52.2 -> 55.5 ms per 8192 permutations: the lanes without a record pile onto one LDS word)
                        if (act_a) {
                            uint32_t pb = Bm[wja] & bja;
                            if (internal) pb |= Bm[wia] & bia;
                            def_a = conf_a || pb != 0u;
                            A[wja] = 0u;
                            if (internal) A[wia] = 0u;
                            if (!def_a) {
                                const uint8_t vi = Xw[ia], vj = X2[ja];
                                Xw[ia] = vj;
                                X2[ja] = vi;
                            } else if (!conf_a && internal) {  // queued behind a poisoned position: its other position is poisoned too
                                atomicOr(&Bm[wja], bja);
                                atomicOr(&Bm[wia], bia);
                            }
                        }
                        if (act_b) {  // after this lane's own record a (had they shared a position, both would be deferred)
                            uint32_t pb = Bm[wjb] & bjb;
                            if (internal) pb |= Bm[wib] & bib;
                            def_b = conf_b || pb != 0u;
                            A[wjb] = 0u;
                            if (internal) A[wib] = 0u;
                            if (!def_b) {
                                const uint8_t vi = Xw[ib], vj = X2[jb];
                                Xw[ib] = vj;
                                X2[jb] = vi;
                            } else if (!conf_b && internal) {
                                atomicOr(&Bm[wjb], bjb);
                                atomicOr(&Bm[wib], bib);
                            }
                        }
                        const uint32_t prio_a = (seq << 11) | L, prio_b = (seq << 11) | (1024u + L);
                        bool sp_a = false, sp_b = false;  // deferred, but the queue was full: the record stays with the lane
                        {
                            const uint64_t ma = __ballot(def_a), mb = __ballot(def_b);
                            if ((ma | mb) != 0ull) {
                                const uint32_t na = (uint32_t)__popcll(ma), nq2 = na + (uint32_t)__popcll(mb);
                                uint32_t base = 0;
                                if ((L & 63u) == 0u) base = atomicAdd(&s_qcount, nq2);
                                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                                if (def_a) {
                                    const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
                                    if (slot < qcap) {
                                        Qrec[slot] = rec_a;
                                        Qprio[slot] = (uint16_t)prio_a;
                                    } else {
                                        sp_a = true;
                                    }
                                }
                                if (def_b) {
                                    const uint32_t slot = base + na + __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
                                    if (slot < qcap) {
                                        Qrec[slot] = rec_b;
                                        Qprio[slot] = (uint16_t)prio_b;
                                    } else {
                                        sp_b = true;
                                    }
                                }
                                if (sp_a || sp_b) s_spill = 1u;
                            }
                        }
                        __syncthreads();
                        PCGQ_ACC(3);
                        const uint32_t qc = s_qcount;
                        const bool spill = s_spill != 0u;
                        ++seq;
                        if (qc == 0u && !spill) {
                            seq = 0;  // nothing is queued: the time index restarts
                        } else if (last_chunk || qc > qcap / 2u || spill || seq == 32u) {
                            // ---- drain: the queued records (lane L: entry L) and the spilled ones in priority rounds
                            const uint32_t nq = min(qc, qcap);
                            PCGQ_CNT(9, 1);
                            PCGQ_CNT(11, qc);
                            const bool was_q = L < nq;
                            const uint32_t rec_q = was_q ? Qrec[L] : 0u;
                            const uint32_t prio_q = was_q ? (uint32_t)Qprio[L] : 0u;
                            const uint32_t jq = rec_q & 0xffffu, iq = rec_q >> 16;
                            const uint32_t hjq = __umulhi(jq * 2654435761u, (uint32_t)PCGQ_SLOTS), hiq = __umulhi(iq * 2654435761u, (uint32_t)PCGQ_SLOTS);
                            const uint32_t hja = __umulhi(ja * 2654435761u, (uint32_t)PCGQ_SLOTS), hia = __umulhi(ia * 2654435761u, (uint32_t)PCGQ_SLOTS);
                            const uint32_t hjb = __umulhi(jb * 2654435761u, (uint32_t)PCGQ_SLOTS), hib = __umulhi(ib * 2654435761u, (uint32_t)PCGQ_SLOTS);
                            bool pend_q = was_q, pend_a = sp_a, pend_b = sp_b;
                            const bool any_sp = spill;  // uniform over the workgroup
                            uint32_t ep = 0;
#if defined(SQGR_PCG_ABLATE) && (SQGR_PCG_ABLATE & 1)  // timing experiment (wrong results): the drain without its rounds
                            if (pend_q) {
                                Bm[jq >> 5] = 0u;
                                if (internal) Bm[iq >> 5] = 0u;
                            }
                            if (pend_a) {
                                Bm[wja] = 0u;
                                if (internal) Bm[wia] = 0u;
                            }
                            if (pend_b) {
                                Bm[wjb] = 0u;
                                if (internal) Bm[wib] = 0u;
                            }
                            pend_q = pend_a = pend_b = false;
#endif
                            // wavefronts without a record (lane L holds queue entry L: all but the first few) only keep the barriers
                            // company — out of the way of the SIMDs' instruction issue
                            if ((L & ~63u) >= nq && !any_sp) {
                                for (;;) {
                                    ++ep;
                                    if ((L & 63u) == 0u) s_wave_any[ep & 1u][L >> 6] = 0u;
                                    __syncthreads();
                                    const uint32_t wa = s_wave_any[ep & 1u][L & (PCGB_THREADS / 64 - 1)];
                                    if (__ballot(wa != 0u) == 0ull) break;
                                }
                            } else
                            for (;;) {
                                ++ep;
                                PCGQ_CNT(10, 1);
                                uint32_t* const T = A + (ep & 1u) * PCGQ_SLOTS;
                                const uint32_t mine_q = (ep << 16) | (0xffffu - prio_q);
                                const uint32_t mine_a = (ep << 16) | (0xffffu - prio_a), mine_b = (ep << 16) | (0xffffu - prio_b);
                                if (pend_q) atomicMax(&T[hjq], mine_q);
                                if (any_sp) {
                                    if (pend_a) atomicMax(&T[hja], mine_a);
                                    if (pend_b) atomicMax(&T[hjb], mine_b);
                                }
                                const bool wave_pending = __ballot(pend_q || pend_a || pend_b) != 0ull;
                                if ((L & 63u) == 0u) s_wave_any[ep & 1u][L >> 6] = wave_pending ? 1u : 0u;
                                __syncthreads();
                                const uint32_t wa = s_wave_any[ep & 1u][L & (PCGB_THREADS / 64 - 1)];
                                uint32_t tq = 0, tiq = 0;
                                if (pend_q) {
                                    tq = T[hjq];
                                    if (internal) tiq = T[hiq];
                                }
                                if (__ballot(wa != 0u) == 0ull) break;  // uniform over the workgroup
                                // a record goes when it is the earliest pending record on its j slot and — window range — no earlier
                                // pending record writes to its i position
                                bool go_q = pend_q && tq == mine_q;
                                if (internal) go_q = go_q && ((tiq >> 16) != ep || (0xffffu - (tiq & 0xffffu)) >= prio_q);
                                if (go_q) {  // (and clears its words of the poison bitmap: all of B's bits belong to drained records)
                                    const uint8_t vi = Xw[iq], vj = X2[jq];
                                    Xw[iq] = vj;
                                    X2[jq] = vi;
                                    Bm[jq >> 5] = 0u;
                                    if (internal) Bm[iq >> 5] = 0u;
                                    pend_q = false;
                                }
                                if (any_sp) {  // (a lane's spilled records are later than every queued one it could share a position with)
                                    uint32_t ta = 0, tia = 0, tb = 0, tib = 0;
                                    if (pend_a) {
                                        ta = T[hja];
                                        if (internal) tia = T[hia];
                                    }
                                    if (pend_b) {
                                        tb = T[hjb];
                                        if (internal) tib = T[hib];
                                    }
                                    bool go_a = pend_a && ta == mine_a, go_b = pend_b && tb == mine_b;
                                    if (internal) {
                                        go_a = go_a && ((tia >> 16) != ep || (0xffffu - (tia & 0xffffu)) >= prio_a);
                                        go_b = go_b && ((tib >> 16) != ep || (0xffffu - (tib & 0xffffu)) >= prio_b);
                                    }
                                    if (go_a) {
                                        const uint8_t vi = Xw[ia], vj = X2[ja];
                                        Xw[ia] = vj;
                                        X2[ja] = vi;
                                        Bm[wja] = 0u;
                                        if (internal) Bm[wia] = 0u;
                                        pend_a = false;
                                    }
                                    if (go_b) {
                                        const uint8_t vi = Xw[ib], vj = X2[jb];
                                        Xw[ib] = vj;
                                        X2[jb] = vi;
                                        Bm[wjb] = 0u;
                                        if (internal) Bm[wib] = 0u;
                                        pend_b = false;
                                    }
                                }
                            }
                            // the tags go back to zero — the region is the claim bitmap again (the last round registered nothing and what a
                            // slow wavefront still reads of it is not used: no barrier in front)
                            for (uint32_t k = L; k < 2u * PCGQ_SLOTS; k += PCGB_THREADS) A[k] = 0u;
                            if (tid == 0) {
                                s_qcount = 0u;
                                s_spill = 0u;
                            }
                            seq = 0;
                            __syncthreads();
                            PCGQ_ACC(4);
                        }
                    };
#pragma unroll
                    for (int t = 0; t < PCGQ_NSUB; ++t) {
                        if (t > 0 && c0 + (uint32_t)t * PCGB_CHUNK_BLOCKS >= nbr) break;  // uniform
                        if (internal)
                            chunk(std::true_type{}, cur[2 * t], cur[2 * t + 1], (cur_hv >> (2 * t)) & 1u, (cur_hv >> (2 * t + 1)) & 1u,
                                  c0 + (uint32_t)(t + 1) * PCGB_CHUNK_BLOCKS >= nbr);
                        else
                            chunk(std::false_type{}, cur[2 * t], cur[2 * t + 1], (cur_hv >> (2 * t)) & 1u, (cur_hv >> (2 * t + 1)) & 1u,
                                  c0 + (uint32_t)(t + 1) * PCGB_CHUNK_BLOCKS >= nbr);
                    }
                    if (last_super && !internal) {
#if !(defined(SQGR_PCG_ABLATE) && (SQGR_PCG_ABLATE & 2))  // timing experiment (wrong results): ranges neither loaded nor stored
                        pcgb_copy_out(row + off + ((size_t)r << logS), Xr, S, tid);
#endif
                        if (tid == 0) s_dirty[r] = 1;
                        __syncthreads();
                        PCGQ_ACC(5);
                    }
                    rr = rr_n;
                    c0 = c0_n;
#pragma unroll
                    for (int q = 0; q < 2 * PCGQ_NSUB; ++q) cur[q] = nxt[q];
                    cur_hv = nxt_hv;
                }
#undef SQGR_PCGQ_PREFETCH
                pcgb_copy_out(row + off + ((size_t)f << logS), Xw, wlen, tid);  // positions of this phase are final
                PCGQ_ACC(6);
            }
        }
        __syncthreads();
    }
}

// rows R[q][row_stride] (bytes) -> columns W[pos * stride + q]
__global__ __launch_bounds__(256) void k_rows_to_columns_u8(int64_t n, int64_t row_stride, const uint8_t* __restrict__ R, int64_t P,
                                                            int64_t stride, uint8_t* __restrict__ W) {
    __shared__ uint8_t tile[64][65];
    const int64_t e0 = (int64_t)blockIdx.x * 64, q0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = (q0 + r < P && e0 + tx < n) ? R[(q0 + r) * row_stride + e0 + tx] : (uint8_t)0;
    __syncthreads();
    // columns P .. stride-1 of the last tile are written too (as label 0): consumers read whole batches of columns and
    // must never see stale bytes there (a stale value >= K would index outside the count kernel's counters)
    for (int r = ty; r < 64; r += 4)
        if (e0 + r < n && q0 + tx < stride) W[(e0 + r) * stride + q0 + tx] = tile[tx][r];
}

// idx[p][i] = W[i][p]  (autocorr row permutations)
__global__ __launch_bounds__(256) void k_columns_to_rows_i32(int64_t n, int64_t stride, const int32_t* __restrict__ W, int64_t P,
                                                             int32_t* __restrict__ idx) {
    __shared__ int32_t tile[32][33];
    const int64_t i0 = (int64_t)blockIdx.x * 32, q0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (i0 + r < n && q0 + tx < P) ? W[(i0 + r) * stride + q0 + tx] : 0;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (q0 + r < P && i0 + tx < n) idx[(size_t)(q0 + r) * n + i0 + tx] = tile[tx][r];
}

}  // namespace sqgr

using namespace sqgr;

// A_k = M^k and G_k = 1 + M + ... + M^(k-1) (mod 2^128) of PCG64's LCG, k = 0..PCGW_TAB-1, as [A_hi, A_lo, G_hi, G_lo]
static int ensure_pcg_jump(DevBuf<uint64_t>& buf) {
    if (buf.p) return SQGR_OK;
    constexpr int TAB = PCGW_TAB2 > PCGW_TAB ? PCGW_TAB2 : PCGW_TAB;  // (k_pcg_draws_bucketed2 jumps up to 64 states ahead)
    std::vector<uint64_t> t((size_t)TAB * 4);
    const unsigned __int128 M = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | (unsigned __int128)0x4385DF649FCCF645ull;
    unsigned __int128 A = 1, G = 0;
    for (int k = 0; k < TAB; ++k) {
        t[4 * k + 0] = (uint64_t)(A >> 64);
        t[4 * k + 1] = (uint64_t)A;
        t[4 * k + 2] = (uint64_t)(G >> 64);
        t[4 * k + 3] = (uint64_t)G;
        G = G * M + 1;
        A = A * M;
    }
    SQGR_TRY(buf.alloc(t.size()));
    SQGR_HIP(hipMemcpy(buf.p, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    return SQGR_OK;
}

static bool pcg_lane_kernel() {  // SQGR_PCG_KERNEL=lane selects the one-thread-per-permutation kernel (comparison runs)
    const char* e = getenv("SQGR_PCG_KERNEL");
    return e && strcmp(e, "lane") == 0;
}
static unsigned pcg_grid(int64_t pc) {  // SQGR_PCG_WAVES caps the number of permutations in flight (tuning runs)
    const char* e = getenv("SQGR_PCG_WAVES");
    const int64_t cap = (e && atoll(e) > 0) ? atoll(e) : pc;
    return (unsigned)std::min<int64_t>(pc, cap);
}
// LDS window (elements, power of two).  Measured on MI355X: a larger window (up to the whole array) buys nothing — the
// trips are bound by their own instruction stream, not by the i-side loads — and costs occupancy.
static uint32_t pcg_window(int64_t n_max) {
    uint32_t ws = 1024;
    while ((int64_t)ws < n_max && ws < 4096u) ws <<= 1;
    return ws;
}
template <typename KernelT>
static int pcg_allow_lds(KernelT kernel, size_t bytes) {
    if (bytes + 16 * 1024 <= 64 * 1024) return SQGR_OK;
    SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SQGR_OK;
}
static int pcg_force_slow() {
    const char* e = getenv("SQGR_PCG_FORCE_SLOW");
    return (e && atoi(e) == 1) ? 1 : 0;
}

namespace sqgr {

// The bucketed replay (k_pcg_draws_bucketed + k_pcg_apply_bucketed) is chosen for long arrays, where the wave kernel's rows fall
// out of every cache; SQGR_PCG_KERNEL=bucket|wave|lane forces a kernel, SQGR_PCG_BUCKET_LOGS the phase length (tests: many phases
// on small arrays).
static int pcg_bucket_logs(int64_t n_lib_max) {
    const char* e = getenv("SQGR_PCG_BUCKET_LOGS");
    int logs = (e && atoi(e) >= 6 && atoi(e) <= 16) ? atoi(e) : 16;
    while (logs < 16 && ceil_div(n_lib_max, (int64_t)1 << logs) > PCGB_MAX_RANGES) ++logs;
    return logs;
}
static bool pcg_use_bucket(int64_t n, int64_t n_lib_max) {
    const char* e = getenv("SQGR_PCG_KERNEL");
    if (e && strcmp(e, "bucket") == 0) return ceil_div(n_lib_max, (int64_t)1 << pcg_bucket_logs(n_lib_max)) <= PCGB_MAX_RANGES;
    if (e && (strcmp(e, "wave") == 0 || strcmp(e, "lane") == 0)) return false;
    // (round 6, tools/pcg_threshold_time.py: with the 128-draw generator and the claims replay the pipeline wins from ~65 000
    // positions on — 1e5 positions x 10 000 permutations 17.5 ms against the wave kernel's 23.9, 80 000: 15.5 / 18.1, 40 000: 4.5 / 4.5)
    return n >= ((int64_t)1 << 16) && ceil_div(n_lib_max, (int64_t)1 << 16) <= PCGB_MAX_RANGES;
}

static int pcg_shuffle_rows_bucketed(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, int64_t n_pad, int n_libs, const uint32_t* lib_off_dev,
                                     const uint8_t* base_pos_dev, const uint64_t* states_dev, int64_t pc, hipStream_t st, const char* timer_name) {
    // geometry from the library sizes (host copy, cached per plan workspace)
    if (ws.lib_off_key != lib_off_dev || (int)ws.lib_off_h.size() != n_libs + 1) {
        ws.lib_off_h.resize((size_t)n_libs + 1);
        SQGR_HIP(hipMemcpyAsync(ws.lib_off_h.data(), lib_off_dev, ((size_t)n_libs + 1) * 4, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
        ws.lib_off_key = lib_off_dev;
        ws.lib_phase_logs = -1;
    }
    int64_t n_lib_max = 1;
    for (int l = 0; l < n_libs; ++l) n_lib_max = std::max<int64_t>(n_lib_max, (int64_t)ws.lib_off_h[l + 1] - ws.lib_off_h[l]);
    PcgBucketGeom geo;
    geo.logS = pcg_bucket_logs(n_lib_max);
    const int64_t S = (int64_t)1 << geo.logS;
    geo.n_ranges = (int)ceil_div(n_lib_max, S);
    geo.bcap = (int)(ceil_div(S, 64) + geo.n_ranges);
    if (ws.lib_phase_logs != geo.logS) {
        std::vector<uint32_t> lp((size_t)n_libs + 1);
        uint32_t acc = 0;
        for (int l = 0; l < n_libs; ++l) {
            lp[l] = acc;
            acc += (uint32_t)ceil_div((int64_t)ws.lib_off_h[l + 1] - ws.lib_off_h[l], S);
        }
        lp[n_libs] = acc;
        SQGR_TRY(ws.lib_phase.ensure(lp.size()));
        SQGR_HIP(hipMemcpyAsync(ws.lib_phase.p, lp.data(), lp.size() * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipStreamSynchronize(st));
        ws.lib_phase_logs = geo.logS;
        ws.phases = (int)acc;
    }
    geo.phases = std::max(ws.phases, 1);

    const size_t rec_words = (size_t)geo.phases * geo.bcap * 64;   // per permutation
    // permutations per pass: <= ~36 GB of records in flight (8192 permutations of 1e6 positions), a quarter of the free memory at most
    size_t free_b = 0, total_b = 0;
    SQGR_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t rec_budget = std::min<size_t>((size_t)36 << 30, free_b / 4);
    const int64_t sub = std::max<int64_t>(64, std::min<int64_t>(pc, (int64_t)(rec_budget / (rec_words * 4))));
    SQGR_TRY(ws.recs.ensure((size_t)sub * rec_words));
    SQGR_TRY(ws.dir.ensure((size_t)sub * geo.phases * geo.bcap));
    SQGR_TRY(ws.nblk.ensure((size_t)sub * geo.phases));
    // generator: 128 draws per trip (default), or rounds 4-5's 64 (SQGR_PCG_DRAWS=64)
    const char* draws_env = getenv("SQGR_PCG_DRAWS");
    const bool draws128 = !(draws_env && atoi(draws_env) == 64);
    size_t lds_g = draws128 ? pcg_draws2_lds_bytes(geo.n_ranges) : (size_t)geo.n_ranges * PCGB_RING * 4;
    if (const char* e = getenv("SQGR_PCG_LDS_PAD")) lds_g += (size_t)atoi(e);  // occupancy experiments
    // conflict tags: what the two windows and the block list leave of the LDS (SQGR_PCG_BUCKET_SLOTS: experiments), at least 3328
    {
        const int64_t room = ((int64_t)160 << 10) - 2 * S - (int64_t)geo.bcap * 4 - 2048;  // (2 KB: the kernel's static LDS)
        int64_t slots = std::max<int64_t>(PCGB_SLOTS, std::min<int64_t>(room / 8, 16384));
        if (const char* e = getenv("SQGR_PCG_BUCKET_SLOTS"))
            if (atoi(e) >= 64) slots = std::min<int64_t>(atoi(e), std::max<int64_t>(room / 8, PCGB_SLOTS));
        geo.slots = (int)slots;
    }
    // replay kernel: exact claims + deferred queue (default), or rounds 4-5's hashed tags (SQGR_PCG_APPLY=tags)
    geo.qcap = PCGQ_CAP;
    if (const char* e = getenv("SQGR_PCG_QUEUE_CAP"))
        if (atoi(e) >= 2 && atoi(e) <= PCGQ_CAP) geo.qcap = atoi(e);
    const char* apply_env = getenv("SQGR_PCG_APPLY");  // (read at every call: tools/pcg_bucket_time.py compares the two)
    const bool use_claims = !(apply_env && strcmp(apply_env, "tags") == 0);
    const size_t lds_a = use_claims ? pcgq_lds_bytes((uint32_t)S, geo.bcap) : 2 * (size_t)S + (size_t)2 * geo.slots * 4 + (size_t)geo.bcap * 4;
    if (draws128) SQGR_TRY(pcg_allow_lds(k_pcg_draws_bucketed2, lds_g));
    else SQGR_TRY(pcg_allow_lds(k_pcg_draws_bucketed, lds_g));
    if (use_claims) SQGR_TRY(pcg_allow_lds(k_pcg_apply_claims, lds_a));
    else SQGR_TRY(pcg_allow_lds(k_pcg_apply_bucketed, lds_a));
    const std::string name_g = std::string(timer_name) + "_draws", name_a = std::string(timer_name) + "_apply";
    const int cus = std::max(ctx->cu_count, 1);
    const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, ((size_t)160 << 10) / (lds_a + 1024)));
    for (int64_t q0 = 0; q0 < pc; q0 += sub) {
        const int64_t qc = std::min(sub, pc - q0);
        SQGR_HIP(hipMemsetAsync(ws.nblk.p, 0, (size_t)qc * geo.phases * 4, st));
        {
            LaunchTimer t(ctx, name_g.c_str(), st);
            if (draws128)
                k_pcg_draws_bucketed2<<<(unsigned)qc, 64, lds_g, st>>>(n_libs, lib_off_dev, ws.lib_phase.p, states_dev + 4 * q0, ws.jump.p, qc, geo,
                                                                       ws.recs.p, ws.dir.p, ws.nblk.p, pcg_force_slow());
            else
                k_pcg_draws_bucketed<<<(unsigned)qc, 64, lds_g, st>>>(n_libs, lib_off_dev, ws.lib_phase.p, states_dev + 4 * q0, ws.jump.p, qc, geo, ws.recs.p,
                                                                      ws.dir.p, ws.nblk.p, pcg_force_slow());
            SQGR_HIP(hipGetLastError());
        }
        {
            LaunchTimer t(ctx, name_a.c_str(), st);
            const unsigned grid = (unsigned)std::min<int64_t>(qc, (int64_t)cus * wg_per_cu);
            if (use_claims) {
#ifdef SQGR_PCG_PROFILE
                unsigned long long zero[16] = {0};
                SQGR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pcgq_prof), zero, sizeof zero));
#endif
                k_pcg_apply_claims<<<grid, PCGB_THREADS, lds_a, st>>>(n_pad, n_libs, lib_off_dev, ws.lib_phase.p, base_pos_dev, qc, geo, ws.recs.p, ws.dir.p,
                                                                      ws.nblk.p, ws.rows.p + (size_t)q0 * n_pad);
#ifdef SQGR_PCG_PROFILE
                unsigned long long prof[16];
                SQGR_HIP(hipStreamSynchronize(st));
                SQGR_HIP(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_pcgq_prof), sizeof prof));
                fprintf(stderr, "pcgq profile (%lld permutations; shader clocks per permutation): phase setup %.0f, range entry %.0f, record wait %.0f, "
                        "record issue %.0f, claim %.0f, apply %.0f, drain %.0f, range out %.0f, window out %.0f | per permutation: chunks %.1f, drains %.1f, "
                        "drain rounds %.1f, queued %.1f\n", (long long)qc, prof[0] / (double)qc, prof[1] / (double)qc, prof[12] / (double)qc,
                        prof[13] / (double)qc, prof[2] / (double)qc, prof[3] / (double)qc, prof[4] / (double)qc, prof[5] / (double)qc, prof[6] / (double)qc,
                        prof[8] / (double)qc, prof[9] / (double)qc, prof[10] / (double)qc, prof[11] / (double)qc);
#endif
            }
            else
                k_pcg_apply_bucketed<<<grid, PCGB_THREADS, lds_a, st>>>(n_pad, n_libs, lib_off_dev, ws.lib_phase.p, base_pos_dev, qc, geo, ws.recs.p, ws.dir.p,
                                                                        ws.nblk.p, ws.rows.p + (size_t)q0 * n_pad);
            SQGR_HIP(hipGetLastError());
        }
    }
    (void)n;
    return SQGR_OK;
}

bool pcg_rows_available() { return !pcg_lane_kernel(); }

int pcg_shuffle_rows(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, int n_libs, const uint32_t* lib_off_dev, const uint8_t* base_pos_dev,
                     const uint64_t* states_dev, int64_t pc, hipStream_t st, const char* timer_name, int64_t* row_stride) {
    if (pcg_lane_kernel()) {
        set_error("SQGR_PCG_KERNEL=lane shuffles columns, not rows");
        return SQGR_ERR_UNSUPPORTED;
    }
    const int64_t n_pad = ceil_div(n, 64) * 64;
    *row_stride = n_pad;
    SQGR_TRY(ensure_pcg_jump(ws.jump));
    SQGR_TRY(ws.rows.ensure((size_t)pc * n_pad));
    // library sizes decide between the wave kernel and the bucketed replay (one library: its size is n)
    int64_t n_lib_max = n;
    if (n_libs > 1 && (ws.lib_off_key != lib_off_dev || (int)ws.lib_off_h.size() != n_libs + 1)) {
        ws.lib_off_h.resize((size_t)n_libs + 1);
        SQGR_HIP(hipMemcpyAsync(ws.lib_off_h.data(), lib_off_dev, ((size_t)n_libs + 1) * 4, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
        ws.lib_off_key = lib_off_dev;
        ws.lib_phase_logs = -1;
    }
    if (n_libs > 1) {
        n_lib_max = 1;
        for (int l = 0; l < n_libs; ++l) n_lib_max = std::max<int64_t>(n_lib_max, (int64_t)ws.lib_off_h[l + 1] - ws.lib_off_h[l]);
    }
    int rc_bucket = SQGR_ERR_UNSUPPORTED;
    if (pcg_use_bucket(n, n_lib_max)) {
        rc_bucket = pcg_shuffle_rows_bucketed(ctx, ws, n, n_pad, n_libs, lib_off_dev, base_pos_dev, states_dev, pc, st, timer_name);
        if (rc_bucket != SQGR_OK && rc_bucket != SQGR_ERR_NOMEM) return rc_bucket;
        // no room for the records (even the floor of 64 permutations per pass: 270 MB at 1e6 positions): the wave kernel below
        // needs none (the call must not fail where it ran before the replay existed)
    }
    if (rc_bucket != SQGR_OK) {
        LaunchTimer t(ctx, timer_name, st);
        const uint32_t wsz = pcg_window(n);
        SQGR_TRY(pcg_allow_lds(k_pcg_shuffle_wave<uint8_t, false>, (size_t)wsz));
        k_pcg_shuffle_wave<uint8_t, false><<<pcg_grid(pc), 64, (size_t)wsz, st>>>(n, n_pad, n_libs, lib_off_dev, base_pos_dev, states_dev,
                                                                                  ws.jump.p, pc, ws.rows.p, pcg_force_slow(), wsz);
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
}

int pcg_shuffle_labels(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, int n_libs, const uint32_t* lib_off_dev, const uint8_t* base_pos_dev,
                       const uint64_t* states_dev, int64_t pc, int64_t stride, uint8_t* W, hipStream_t st, const char* timer_name) {
    if (pcg_lane_kernel()) {
        LaunchTimer t(ctx, timer_name, st);
        const int64_t pc64 = std::min<int64_t>(stride, ceil_div(pc, 64) * 64);
        if (pc64 > pc)  // columns past pc that whole-batch consumers still read: label 0 instead of stale bytes
            SQGR_HIP(hipMemset2DAsync(W + pc, (size_t)stride, 0, (size_t)(pc64 - pc), (size_t)n, st));
        k_pcg_shuffle<uint8_t, false><<<(unsigned)ceil_div(pc, 64), 64, 0, st>>>(n, n_libs, lib_off_dev, base_pos_dev, states_dev, pc, stride, W);
        SQGR_HIP(hipGetLastError());
        return SQGR_OK;
    }
    int64_t n_pad = 0;
    SQGR_TRY(pcg_shuffle_rows(ctx, ws, n, n_libs, lib_off_dev, base_pos_dev, states_dev, pc, st, timer_name, &n_pad));
    {
        const std::string name_t = std::string(timer_name) + "_rows_to_columns";
        LaunchTimer t(ctx, name_t.c_str(), st);
        k_rows_to_columns_u8<<<dim3((unsigned)ceil_div(n, 64), (unsigned)ceil_div(pc, 64)), 256, 0, st>>>(n, n_pad, ws.rows.p, pc, stride, W);
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
}

int pcg_permutations_dev(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, const uint64_t* states_dev, int64_t pc, int32_t* idx_dev,
                         hipStream_t st) {
    if (!ws.span.p) {
        SQGR_TRY(ws.span.alloc(2));
        const uint32_t span_h[2] = {0u, (uint32_t)n};
        SQGR_HIP(hipMemcpy(ws.span.p, span_h, 8, hipMemcpyHostToDevice));
    }
    LaunchTimer t(ctx, "autocorr_pcg64_permutation", st);
    if (pcg_lane_kernel()) {
        const int64_t stride = ceil_div(pc, 64) * 64;
        SQGR_TRY(ws.cols.ensure((size_t)n * stride));
        k_pcg_shuffle<int32_t, true><<<(unsigned)ceil_div(pc, 64), 64, 0, st>>>(n, 1, ws.span.p, nullptr, states_dev, pc, stride, ws.cols.p);
        k_columns_to_rows_i32<<<dim3((unsigned)ceil_div(n, 32), (unsigned)ceil_div(pc, 32)), 256, 0, st>>>(n, stride, ws.cols.p, pc, idx_dev);
    } else {  // rows are the wanted output already: idx[q][i]
        SQGR_TRY(ensure_pcg_jump(ws.jump));
        const uint32_t wsz = pcg_window(n);
        SQGR_TRY(pcg_allow_lds(k_pcg_shuffle_wave<int32_t, true>, (size_t)wsz * 4));
        k_pcg_shuffle_wave<int32_t, true><<<pcg_grid(pc), 64, (size_t)wsz * 4, st>>>(n, n, 1, ws.span.p, nullptr, states_dev, ws.jump.p, pc,
                                                                                     idx_dev, pcg_force_slow(), wsz);
    }
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

}  // namespace sqgr

extern "C" {

int sqgr_pcg64_permutations(sqgr_ctx* ctx, int64_t n, const uint64_t* pcg_states, int64_t n_perms, int32_t* out_idx) {
    SQGR_REQUIRE(ctx && pcg_states && out_idx && n > 0 && n < (int64_t)0x7fffffff && n_perms >= 0, "bad argument");
    if (n_perms == 0) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(ceil_div(n_perms, 64) * 64, (((int64_t)8 << 30) / (n * 4)) / 64 * 64));
    PcgWorkspace ws;
    DevBuf<int32_t> idx;
    DevBuf<uint64_t> states;
    SQGR_TRY(idx.alloc((size_t)chunk * n));
    SQGR_TRY(states.alloc((size_t)chunk * 4));
    for (int64_t c0 = 0; c0 < n_perms; c0 += chunk) {
        const int64_t pc = std::min(chunk, n_perms - c0);
        SQGR_HIP(hipMemcpyAsync(states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
        SQGR_TRY(pcg_permutations_dev(ctx, ws, n, states.p, pc, idx.p, st));
        SQGR_HIP(hipMemcpyAsync(out_idx + (size_t)c0 * n, idx.p, (size_t)pc * n * 4, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
    }
    return SQGR_OK;
}

}  // extern "C"
