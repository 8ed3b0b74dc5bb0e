// libsqgr: neighbourhood-enrichment permutation test on the device-resident CSR graph.
//
// Reference semantics (scverse/squidpy, src/squidpy/gr/_nhood.py):
//   :54-141   generated numba kernel  count[a,b] = sum_{i: lab_i=a} #{j in N(i): lab_j=b}  (uint32)
//   :516-547  per permutation: shuffle labels (per library: gr/_utils.py:185-213), count again
//   :231      zscore = (count - mean_p) / std_p
//
// MI355X design: B (16 or 32) permutations are processed per pass over the graph.
//   k_shuffle : thread per spot; B keyed Feistel bijections -> shuffled labels written as ONE B-byte
//               row per spot: slab[spot][b] (uint8).  One 16/32-byte gather later serves B permutations.
//   k_count   : edge-parallel over the COO view (erow, indices); 4 lanes per edge, each lane owns B/4
//               permutations; K*K*B counters live in LDS, laid out [pair][b] so that the 32 lanes of a
//               DS lane group hit distinct banks (B=32) — ds_add_u32 without bank conflicts; block-local
//               histograms are written out as partials (plain coalesced stores, no global atomics).
//   k_reduce  : thread per (pair, b): sums the partials over blocks -> exact per-permutation count,
//               accumulates d=count-shift and d*d in 64-bit integers (order independent, deterministic).
// Variants measured on MI355X and dropped (round 3, commit 652982d holds their code; profiles/r03_nhood_experiments.json):
// a deferred exact route of k_shuffle (flagged words appended to lists by returning atomics and recomputed by a second
// kernel: 1.19 -> 1.62 ms per launch, and skipping the route altogether would only save 5 %); k_count with two lanes per
// label row (global_load_dwordx2, half the gather instructions: bit-exact, 1.11 -> 1.56 ms — 8- and 16-byte-per-lane
// gathers cost 16 clk per wave instruction in the address/L1 path whatever they touch, tools/ubench_count_shape.hip);
// k_shuffle and k_count of consecutive launch groups sharing the CUs on two streams (no gain with 1 or 2 count blocks per CU).
#include "sqgr_common.h"
#include "sqgr_rng.h"
#include "sqgr_shuffle.h"
#include "sqgr_pcg.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace sqgr {

constexpr int COUNT_THREADS = 1024;
constexpr size_t LDS_BUDGET = 160 * 1024;

// ---------------------------------------------------------------------------------------------- key generation
// Keys of one slab row (B permutations perm_row .. perm_row + B - 1, perm_row a multiple of 16) — see sqgr_rng.h for the
// two-level construction.  Layout in 32-bit words, every word two packed 16-bit lanes (two permutations per packed-16
// instruction of the label shuffle):
//   group keys : [g][lib][8]   g < B/16: the 8 round keys of group perm_row/16 + g in both halves
//   sigma keys : [t][lib][2]   t < B/2 : the 2 round keys of permutations perm_row + 2t (low half) and + 2t + 1 (high half)
// (a row of 16 never takes fewer than 64 words: the independent-bijection variant of the generator, k_shuffle_indep, keeps the 8
// round keys of each of its 8 permutation pairs there)
__host__ __device__ constexpr int key_words_per_row(int B, int n_libs) {
    return n_libs * ((B / FEISTEL_GROUP) * 8 + (B / 2) * 2) < (B / 2) * 8 ? (B / 2) * 8 : n_libs * ((B / FEISTEL_GROUP) * 8 + (B / 2) * 2);
}

__global__ void k_keygen(uint64_t seed, int64_t perm0, int nrows, int B, int n_libs, uint32_t* __restrict__ keys) {
    const int items = B / FEISTEL_GROUP + B / 2;  // per (row, library)
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)nrows * n_libs * items) return;
    const int item = (int)(t % items);
    const uint32_t lib = (uint32_t)((t / items) % n_libs);
    const int64_t row = t / ((int64_t)items * n_libs);
    uint32_t* out = keys + row * key_words_per_row(B, n_libs);
    const int64_t perm_row = perm0 + row * B;
    if (item < B / FEISTEL_GROUP) {
        uint32_t rk[8];
        group_keys(seed, (uint64_t)(perm_row / FEISTEL_GROUP + item), lib, rk);
#pragma unroll
        for (int i = 0; i < 8; ++i) out[((size_t)item * n_libs + lib) * 8 + i] = (rk[i] & 0xFFFFu) * 0x10001u;
    } else {
        const int tt = item - B / FEISTEL_GROUP;
        uint32_t ra[2], rb[2];
        sigma_keys(seed, (uint64_t)(perm_row + 2 * tt), lib, ra);
        sigma_keys(seed, (uint64_t)(perm_row + 2 * tt + 1), lib, rb);
        uint32_t* o = out + (size_t)(B / FEISTEL_GROUP) * n_libs * 8 + ((size_t)tt * n_libs + lib) * 2;
        o[0] = (ra[0] & 0xFFFFu) | (rb[0] << 16);
        o[1] = (ra[1] & 0xFFFFu) | (rb[1] << 16);
    }
}

// keys of the INDEPENDENT variant (SQGR_SHUFFLE_INDEPENDENT=1, bench.py's `nhood_independent_bijections` leg): every permutation
// its own 8-round bijection, keyed like a row permutation of spatial_autocorr (round_keys: Philox counter words j = 0, 1) —
// [pair t < 8][round r < 8] per row of 16, permutation perm_row + 2t in the low half, + 2t + 1 in the high half
__global__ void k_keygen_indep(uint64_t seed, int64_t perm0, int nrows, uint32_t* __restrict__ keys) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)nrows * 8) return;
    const int64_t row = t / 8;
    const int pair = (int)(t % 8);
    uint32_t ra[8], rb[8];
    round_keys(seed, (uint64_t)(perm0 + row * 16 + 2 * pair), 0u, ra);
    round_keys(seed, (uint64_t)(perm0 + row * 16 + 2 * pair + 1), 0u, rb);
    uint32_t* out = keys + row * key_words_per_row(16, 1) + pair * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) out[r] = (ra[r] & 0xFFFFu) | (rb[r] << 16);
}

// ---------------------------------------------------------------------------------------------- label shuffle
// A uniformly random arrangement of the label multiset does not depend on the order of the base vector, so
// the base is taken *sorted by label* (inside each library): the label at sorted rank x is
//   #{k >= 1 : cum[k] <= x},   cum[k] = number of spots with label < k   (cum[K] = UINT_MAX sentinel).
// No memory gather: rank x = a*B + b arrives as its two digits; one LDS word per high digit a (block table, see
// nhood_build) holds the label of the block's first rank (byte 0) and the low digit where the next label starts
// (bits 16-31, 0xFFFF: none), so the label is `lab0 + (b >= next)`.  The value K — one past the last label — is the
// SENTINEL: blocks the two-field form cannot describe (several label starts, skipped empty categories) carry lab0 = K,
// and in the block that reaches past the library's last rank "next" is the first low digit outside [0, n) when that
// block ends in label K-1 (so the cycle-walking case yields K as well).  Sentinel bytes are rare (ranks >= n: fewer than
// 1 in B') and take the exact route afterwards: re-walk where x >= n, then x against the boundary table.
//   slab[(batch*n + i)*B + b] = label_at_rank( sigma_p( pi_g( rank_i ) ) ),  p = perm_row + b, g = p / 16
// One evaluation of the 8-round group bijection per spot and 16 permutations; per permutation the 2-round sigma network
// (two permutations per packed-16 instruction) and the table look-up: the compare reads its operands as 16-bit halves
// in place and the add-with-carry writes the label straight into its byte of the output word (SDWA) — two VALU
// instructions per label after the LDS read.
struct LibDom {
    FeistelDomain dom;
    uint32_t aoff;  // offset of this library's block table
};

// label byte J of `word` <- e.byte0 + (b.half H >= e.half1)
template <int J, int H>
__device__ __forceinline__ void put_label(uint32_t& word, uint32_t e, uint32_t bpk, uint32_t zero) {
#define SQGR_PUT(JS, HS)                                                                                                      \
    asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:WORD_" HS "\n\t"                                              \
        "v_addc_co_u32_sdwa %0, vcc, %1, %3, vcc dst_sel:BYTE_" JS " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n\t" \
        "s_nop 0"                                                                                                            \
        : "+v"(word)                                                                                                         \
        : "v"(e), "v"(bpk), "v"(zero)                                                                                        \
        : "vcc")
    if constexpr (J == 0 && H == 0) SQGR_PUT("0", "0");
    else if constexpr (J == 1 && H == 1) SQGR_PUT("1", "1");
    else if constexpr (J == 2 && H == 0) SQGR_PUT("2", "0");
    else if constexpr (J == 3 && H == 1) SQGR_PUT("3", "1");
    else static_assert(J < 0, "byte J of the word holds the permutation of packed half J & 1");
#undef SQGR_PUT
}

// The 16 shuffled labels of spot i (4 words, label b in byte b & 3 of word b >> 2) into a batch's 16 * n bytes of the slab.
// pw = 16: row i of [n][16] — what k_count gathers.  pw = 8 | 4 | 2 | 1 (51 <= K <= 202 clusters, k_count_pass): 16 / pw PLANES
// [n][pw], plane q = permutations [q * pw, (q + 1) * pw) — a pass of pw permutations then gathers from dense rows of exactly
// the bytes it uses (round 5: out of 16-byte rows a pass of 4 pulled four times the cache lines through L1 and L2).
__device__ __forceinline__ void slab_store16(uint8_t* __restrict__ batch_base, int64_t n, int64_t i, int pw, uint32_t w0, uint32_t w1,
                                             uint32_t w2, uint32_t w3) {
    if (pw == 16) {
        *reinterpret_cast<uint4*>(batch_base + (size_t)i * 16) = make_uint4(w0, w1, w2, w3);
    } else if (pw == 8) {
        *reinterpret_cast<uint2*>(batch_base + (size_t)i * 8) = make_uint2(w0, w1);
        *reinterpret_cast<uint2*>(batch_base + (size_t)n * 8 + (size_t)i * 8) = make_uint2(w2, w3);
    } else if (pw == 4) {
        uint32_t* d = reinterpret_cast<uint32_t*>(batch_base) + i;
        d[0] = w0; d[(size_t)n] = w1; d[(size_t)2 * n] = w2; d[(size_t)3 * n] = w3;
    } else if (pw == 2) {
        uint16_t* d = reinterpret_cast<uint16_t*>(batch_base) + i;
        const uint32_t w[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int q = 0; q < 8; ++q) d[(size_t)q * n] = (uint16_t)(w[q >> 1] >> (16 * (q & 1)));
    } else {
        uint8_t* d = batch_base + i;
        const uint32_t w[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int q = 0; q < 16; ++q) d[(size_t)q * n] = (uint8_t)(w[q >> 2] >> (8 * (q & 3)));
    }
}

template <int B, bool HAS_LIBS, bool SMALLK>
__global__ __launch_bounds__(256) void k_shuffle(int64_t n, const uint32_t* __restrict__ cum, int kpad, int blk_words, int K,
                                                 const uint32_t* __restrict__ keys, LibDom dom0, int n_libs, int nrows,
                                                 const int32_t* __restrict__ lib_of, const int32_t* __restrict__ rank_of,
                                                 const LibDom* __restrict__ libdoms, uint8_t* __restrict__ slab_all, int pw,
                                                 const int32_t* __restrict__ spot_of) {
    // spot_of != NULL (no libraries): the plan lives on a RENUMBERED twin of the caller's graph (sqgr_graph_renumbered) — slab row i
    // belongs to the caller's observation spot_of[i], whose rank the generator permutes: the labels are the ones the plan on the
    // caller's own graph gives that observation, stored where the renumbered edge lists look for them
    extern __shared__ uint32_t s_lds[];               // [blk_words] block table (byte offset 0), then [n_libs][kpad] boundaries
    uint32_t* s_cum = s_lds + blk_words;
    for (int t = threadIdx.x; t < n_libs * kpad; t += 256) s_cum[t] = cum[t];
    for (int t = threadIdx.x; t < blk_words; t += 256) s_lds[t] = cum[n_libs * kpad + t];
    __syncthreads();
    // A thread evaluates TWO groups of 16 permutations per spot, one in each 16-bit half of the packed registers, so the
    // 8-round group bijection costs one packed evaluation per 32 labels: the two groups of a 32-permutation row, or the
    // groups of two consecutive 16-permutation rows (blockIdx.y counts row pairs then).
    constexpr int NG = B / FEISTEL_GROUP;
    const int row0 = (NG == 2) ? blockIdx.y : 2 * blockIdx.y;
    const int row1 = (NG == 2) ? row0 : min(row0 + 1, nrows - 1);  // an odd last row is paired with itself (stored once)
    const bool store1 = (NG == 2) || row0 + 1 < nrows;
    const size_t row_words = key_words_per_row(B, n_libs);
    const uint32_t* kgA = keys + (size_t)row0 * row_words;                                   // group keys of group A ...
    const uint32_t* kgB = (NG == 2) ? kgA + (size_t)n_libs * 8 : keys + (size_t)row1 * row_words;  // ... and of group B
    const uint32_t* ksA = keys + (size_t)row0 * row_words + (size_t)NG * n_libs * 8;         // sigma keys of group A's 8 pairs
    const uint32_t* ksB = (NG == 2) ? ksA + (size_t)(FEISTEL_GROUP / 2) * n_libs * 2
                                    : keys + (size_t)row1 * row_words + (size_t)NG * n_libs * 8;
    const uint32_t zero = 0;
    // any label byte >= K ?  K <= 126: bytes are < 128, adding 128 - K sets bit 7 exactly for the sentinels
    const uint32_t sent_add = (uint32_t)(128 - K) * 0x01010101u;
    // grid-stride over spots (launch_shuffle_raw caps the grid)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    LibDom ld = dom0;
    uint32_t x0 = (uint32_t)i, lib = 0;
    if (HAS_LIBS) {
        lib = (uint32_t)lib_of[i];
        ld = libdoms[lib];
        x0 = (uint32_t)rank_of[i];
    } else if (spot_of) {
        x0 = (uint32_t)spot_of[i];
    }
    const FeistelDomain dom = ld.dom;
    const uint32_t* tab = s_cum + lib * kpad + 1;  // tab[k] = cum[k + 1]
    // The block table sits at LDS byte offset 0 (this kernel has no static LDS, the dynamic segment starts at 0): reading it
    // through an explicit address-space-3 address spares the add of the segment base the compiler otherwise emits per look-up.
    typedef __attribute__((address_space(3))) const uint32_t lds_word;
    const uint32_t blk_base = HAS_LIBS ? ld.aoff * 4u : 0u;
    auto blk_at = [&](uint32_t byte_off) { return *reinterpret_cast<lds_word*>((uintptr_t)(blk_base + byte_off)); };
    const uint32_t a0 = x0 / dom.B, b0 = x0 - a0 * dom.B;  // one division per spot, shared by all permutations

    // pi_gA (low halves) and pi_gB (high halves) in one packed evaluation; cycle walk per half (rare: < 1/B of the ranks)
    u16x2 ga = (u16x2)((unsigned short)a0), gb = (u16x2)((unsigned short)b0);
    {
        uint32_t kp[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) kp[r] = (kgA[(size_t)lib * 8 + r] & 0xFFFFu) | (kgB[(size_t)lib * 8 + r] & 0xFFFF0000u);
        const uint32_t* const pg[1] = {kp};
        bool need0 = true, need1 = true;
        do {
            u16x2 na[1] = {ga}, nb[1] = {gb};
            feistel_rounds<1>(na, nb, dom, pg);
            if (need0) { ga.x = na[0].x; gb.x = nb[0].x; }
            if (need1) { ga.y = na[0].y; gb.y = nb[0].y; }
            need0 = __umul24((uint32_t)ga.x, dom.B) + (uint32_t)gb.x >= dom.n;
            need1 = __umul24((uint32_t)ga.y, dom.B) + (uint32_t)gb.y >= dom.n;
        } while (need0 | need1);
    }
    // the 16 labels of one group: sigma_p on the group's image (in both halves), two permutations per packed evaluation
    auto emit_group = [&](const u16x2 gsa, const u16x2 gsb, const uint32_t* ks, uint32_t (&out)[4]) {
#pragma unroll
        for (int w = 0; w < FEISTEL_GROUP / 4; ++w) {
            uint32_t word = 0;
            uint32_t apk[2], bpk[2];  // sigma images of the word's two pairs (kept for the exact route)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const uint32_t* const pk[1] = {ks + ((size_t)(w * 2 + jj) * n_libs + lib) * 2};
                u16x2 a[1] = {gsa}, b[1] = {gsb};
                sigma_rounds<1>(a, b, dom, pk);
                apk[jj] = __builtin_bit_cast(uint32_t, a[0]);
                bpk[jj] = __builtin_bit_cast(uint32_t, b[0]);
                const uint32_t a4 = apk[jj] << 2;  // both halves at once (a < 2^14): byte offsets into the block table
                const uint32_t e0 = blk_at(a4 & 0xFFFFu);
                const uint32_t e1 = blk_at(a4 >> 16);
                if (jj == 0) {
                    put_label<0, 0>(word, e0, bpk[jj], zero);
                    put_label<1, 1>(word, e1, bpk[jj], zero);
                } else {
                    put_label<2, 0>(word, e0, bpk[jj], zero);
                    put_label<3, 1>(word, e1, bpk[jj], zero);
                }
            }
            bool sentinel;
            if constexpr (SMALLK) {
                sentinel = ((word + sent_add) & 0x80808080u) != 0;
            } else {  // K = 256 has no sentinel value left in a byte: every label takes the exact route
                const uint32_t k = (uint32_t)K;
                sentinel = k > 255u || (word & 0xFFu) >= k || ((word >> 8) & 0xFFu) >= k || ((word >> 16) & 0xFFu) >= k || (word >> 24) >= k;
            }
            if (sentinel) {  // exact route (rare): re-walk sigma where the image left [0, n), then rank against the boundaries
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (K <= 255 && ((word >> (8 * j)) & 0xFFu) < (uint32_t)K) continue;
                    const int jj = j >> 1, sh = (j & 1) * 16;
                    const uint32_t* sk = ks + ((size_t)(w * 2 + jj) * n_libs + lib) * 2;
                    const uint32_t k0 = (sk[0] >> sh) & 0xFFFFu, k1 = (sk[1] >> sh) & 0xFFFFu;
                    uint32_t a = (apk[jj] >> sh) & 0xFFFFu, b = (bpk[jj] >> sh) & 0xFFFFu;
                    uint32_t x = a * dom.B + b;
                    while (x >= dom.n) {
                        uint32_t t = b + feistel_F1(a, k0, dom.bsh);
                        t = t >= dom.B ? t - dom.B : t;
                        b = t >= dom.B ? t - dom.B : t;
                        a = (a + feistel_F1(b, k1, dom.ash)) & (dom.A - 1u);
                        x = a * dom.B + b;
                    }
                    const uint32_t e = blk_at(a * 4u);  // the table again: nearly always enough
                    uint32_t l = (e & 0xFFFFu) + (b >= (e >> 16) ? 1u : 0u);
                    if (l >= (uint32_t)K) {  // a block the two-field form cannot describe: rank against the boundaries, from the
                        l = K <= 255 ? (e >> 8) & 0xFFu : 0u;  // block's first label on (skewed cluster sizes: 1-3 steps instead of ~K/2)
                        while (x >= tab[l]) ++l;  // sentinel UINT_MAX stops it
                    }
                    word = (word & ~(0xFFu << (8 * j))) | (l << (8 * j));
                }
            }
            out[w] = word;
        }
    };
    uint32_t outA[4], outB[4];
    emit_group((u16x2)(ga.x), (u16x2)(gb.x), ksA, outA);
    if (store1) emit_group((u16x2)(ga.y), (u16x2)(gb.y), ksB, outB);
    if constexpr (NG == 2) {
        uint4* dst = reinterpret_cast<uint4*>(slab_all + ((size_t)row0 * n + i) * B);
        dst[0] = make_uint4(outA[0], outA[1], outA[2], outA[3]);
        dst[1] = make_uint4(outB[0], outB[1], outB[2], outB[3]);
    } else {
        slab_store16(slab_all + (size_t)row0 * n * 16, n, i, pw, outA[0], outA[1], outA[2], outA[3]);
        if (store1) slab_store16(slab_all + (size_t)row1 * n * 16, n, i, pw, outB[0], outB[1], outB[2], outB[3]);
    }
    }
}

// The generator WITHOUT its shortcut (what do 16 independent bijections cost?): slab label b of row `row` =
// label_at_rank(pi_p(rank_i)) with pi_p the full 8-round bijection of permutation p = perm_row + b itself — no group bijection
// shared by 16 permutations, no 2-round sigma network.  Two permutations per packed evaluation, 8 evaluations of 8 rounds per spot
// and row (the two-level kernel: half an evaluation of 8 rounds + 8 of 2).  Same label look-up (block table, sentinel, exact
// route), same slab layout; no libraries.  Restated in oracle/devrng.py (independent_label_permutations).
template <bool SMALLK>
__global__ __launch_bounds__(256) void k_shuffle_indep(int64_t n, const uint32_t* __restrict__ cum, int kpad, int blk_words, int K,
                                                       const uint32_t* __restrict__ keys, LibDom dom0, uint8_t* __restrict__ slab_all, int pw) {
    extern __shared__ uint32_t s_lds[];
    uint32_t* s_cum = s_lds + blk_words;
    for (int t = threadIdx.x; t < kpad; t += 256) s_cum[t] = cum[t];
    for (int t = threadIdx.x; t < blk_words; t += 256) s_lds[t] = cum[kpad + t];
    __syncthreads();
    const int row = blockIdx.y;
    const uint32_t* krow = keys + (size_t)row * key_words_per_row(16, 1);
    const uint32_t zero = 0;
    const uint32_t sent_add = (uint32_t)(128 - K) * 0x01010101u;
    const FeistelDomain dom = dom0.dom;
    const uint32_t* tab = s_cum + 1;
    typedef __attribute__((address_space(3))) const uint32_t lds_word;
    auto blk_at = [&](uint32_t byte_off) { return *reinterpret_cast<lds_word*>((uintptr_t)byte_off); };
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t x0 = (uint32_t)i;
        const uint32_t a0 = x0 / dom.B, b0 = x0 - a0 * dom.B;
        uint32_t out[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t word = 0, apk[2], bpk[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const uint32_t* const pk[1] = {krow + (w * 2 + jj) * 8};
                u16x2 ga = (u16x2)((unsigned short)a0), gb = (u16x2)((unsigned short)b0);
                bool need0 = true, need1 = true;
                do {  // the bijection, cycle-walked per half
                    u16x2 na[1] = {ga}, nb[1] = {gb};
                    feistel_rounds<1>(na, nb, dom, pk);
                    if (need0) { ga.x = na[0].x; gb.x = nb[0].x; }
                    if (need1) { ga.y = na[0].y; gb.y = nb[0].y; }
                    need0 = __umul24((uint32_t)ga.x, dom.B) + (uint32_t)gb.x >= dom.n;
                    need1 = __umul24((uint32_t)ga.y, dom.B) + (uint32_t)gb.y >= dom.n;
                } while (need0 | need1);
                apk[jj] = __builtin_bit_cast(uint32_t, ga);
                bpk[jj] = __builtin_bit_cast(uint32_t, gb);
                const uint32_t a4 = apk[jj] << 2;
                const uint32_t e0 = blk_at(a4 & 0xFFFFu);
                const uint32_t e1 = blk_at(a4 >> 16);
                if (jj == 0) {
                    put_label<0, 0>(word, e0, bpk[jj], zero);
                    put_label<1, 1>(word, e1, bpk[jj], zero);
                } else {
                    put_label<2, 0>(word, e0, bpk[jj], zero);
                    put_label<3, 1>(word, e1, bpk[jj], zero);
                }
            }
            bool sentinel;
            if constexpr (SMALLK) {
                sentinel = ((word + sent_add) & 0x80808080u) != 0;
            } else {
                const uint32_t k = (uint32_t)K;
                sentinel = k > 255u || (word & 0xFFu) >= k || ((word >> 8) & 0xFFu) >= k || ((word >> 16) & 0xFFu) >= k || (word >> 24) >= k;
            }
            if (sentinel) {  // a block the two-field table form cannot describe: rank against the boundaries (the image is < n already)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (K <= 255 && ((word >> (8 * j)) & 0xFFu) < (uint32_t)K) continue;
                    const int jj = j >> 1, sh = (j & 1) * 16;
                    const uint32_t a = (apk[jj] >> sh) & 0xFFFFu, b = (bpk[jj] >> sh) & 0xFFFFu;
                    const uint32_t x = a * dom.B + b;
                    const uint32_t e = blk_at(a * 4u);
                    uint32_t l = (e & 0xFFFFu) + (b >= (e >> 16) ? 1u : 0u);
                    if (l >= (uint32_t)K) {
                        l = K <= 255 ? (e >> 8) & 0xFFu : 0u;  // (the block's first label: k_shuffle)
                        while (x >= tab[l]) ++l;
                    }
                    word = (word & ~(0xFFu << (8 * j))) | (l << (8 * j));
                }
            }
            out[w] = word;
        }
        slab_store16(slab_all + (size_t)row * n * 16, n, i, pw, out[0], out[1], out[2], out[3]);
    }
}

// More than 256 clusters: the same two-level generator writing 16-bit labels, slab16[(batch*n + i)*16 + b] (32-byte rows).
// Plain arithmetic on the unpacked digits; blocks the two-field table form cannot describe are frequent here (many label
// starts per block), so the exact route ranks x against the boundary table by binary search.
template <bool HAS_LIBS>
__global__ __launch_bounds__(256) void k_shuffle16(int64_t n, const uint32_t* __restrict__ cum, int kpad, int blk_words, int K,
                                                   const uint32_t* __restrict__ keys, LibDom dom0, int n_libs,
                                                   const int32_t* __restrict__ lib_of, const int32_t* __restrict__ rank_of,
                                                   const LibDom* __restrict__ libdoms, uint16_t* __restrict__ slab_all, int tab_lds) {
    extern __shared__ uint32_t s_lds[];
    uint32_t* s_cum = s_lds + blk_words;
    if (tab_lds)  // (else: the boundaries stay in global memory — sqgr_nhood::tab_lds)
        for (int t = threadIdx.x; t < n_libs * kpad; t += 256) s_cum[t] = cum[t];
    for (int t = threadIdx.x; t < blk_words; t += 256) s_lds[t] = cum[n_libs * kpad + t];
    __syncthreads();
    constexpr int B = 16;
    const int batch = blockIdx.y;
    const uint32_t* kg = keys + (size_t)batch * key_words_per_row(B, n_libs);
    const uint32_t* ks = kg + (size_t)n_libs * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        LibDom ld = dom0;
        uint32_t x0 = (uint32_t)i, lib = 0;
        if (HAS_LIBS) {
            lib = (uint32_t)lib_of[i];
            ld = libdoms[lib];
            x0 = (uint32_t)rank_of[i];
        }
        const FeistelDomain dom = ld.dom;
        const uint32_t* tab = s_cum + lib * kpad + 1;  // tab[k] = cum[k + 1]
        const uint32_t* gtab = cum + lib * kpad + 1;   // the same table where it lies in global memory (tab_lds == 0)
        const uint32_t* blk = s_lds + ld.aoff;
        uint32_t ga = x0 / dom.B, gb = x0 - ga * dom.B;
        const uint32_t* gk = kg + (size_t)lib * 8;
        do {  // pi_g (low halves of the packed key words)
            for (int r = 0; r < FEISTEL_ROUNDS; r += 2) {
                ga = (ga + feistel_F1(gb, gk[r], dom.ash)) & (dom.A - 1u);
                uint32_t t = gb + feistel_F1(ga, gk[r + 1], dom.bsh);
                t = t >= dom.B ? t - dom.B : t;
                gb = t >= dom.B ? t - dom.B : t;
            }
        } while (ga * dom.B + gb >= dom.n);
        uint32_t out[B / 2];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t* sk = ks + ((size_t)(j >> 1) * n_libs + lib) * 2;
            const int sh = (j & 1) * 16;
            const uint32_t k0 = (sk[0] >> sh) & 0xFFFFu, k1 = (sk[1] >> sh) & 0xFFFFu;
            uint32_t a = ga, b = gb, x;
            do {  // sigma_p, cycle-walked
                uint32_t t = b + feistel_F1(a, k0, dom.bsh);
                t = t >= dom.B ? t - dom.B : t;
                b = t >= dom.B ? t - dom.B : t;
                a = (a + feistel_F1(b, k1, dom.ash)) & (dom.A - 1u);
                x = a * dom.B + b;
            } while (x >= dom.n);
            const uint32_t e = blk[a];
            uint32_t l = (e & 0xFFFFu) + (b >= (e >> 16) ? 1u : 0u);
            if (l >= (uint32_t)K) {  // largest l with cum[l] <= x  <=>  first l with x < tab[l]
                uint32_t lo = 0, hi = (uint32_t)K - 1u;
                if (tab_lds) {
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (x < tab[mid]) hi = mid; else lo = mid + 1;
                    }
                } else {
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (x < gtab[mid]) hi = mid; else lo = mid + 1;
                    }
                }
                l = lo;
            }
            if (j & 1) out[j >> 1] |= l << 16; else out[j >> 1] = l;
        }
        uint4* dst = reinterpret_cast<uint4*>(slab_all + ((size_t)batch * n + i) * B);
        dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
        dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    }
}

// numpy streams with 16-bit labels: Generator.shuffle(x) leaves x[perm] behind, perm = the same generator's permutation(n)
// (both run the same Fisher-Yates on the same draws): slab16[(batch*n + i)*16 + b] = base16[idx[p][i]], p = p0 + batch*16 + b
__global__ __launch_bounds__(256) void k_gather_labels16(int64_t n, const uint16_t* __restrict__ base16, const int32_t* __restrict__ idx,
                                                         int64_t p0, int64_t p_valid, uint16_t* __restrict__ slab_all) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int batch = blockIdx.y;
    uint32_t out[8];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int64_t p = p0 + (int64_t)batch * 16 + b;
        const uint32_t l = p < p_valid ? base16[idx[(size_t)p * n + i]] : 0u;
        if (b & 1) out[b >> 1] |= l << 16; else out[b >> 1] = l;
    }
    uint4* dst = reinterpret_cast<uint4*>(slab_all + ((size_t)batch * n + i) * 16);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

// counts for 16-bit labels: K*K counters per permutation do not fit LDS -> device-scope atomics into partial[batch][pair*16 + b]
__global__ __launch_bounds__(256) void k_count_wide16(int64_t nnz, const int32_t* __restrict__ erow, const int32_t* __restrict__ indices,
                                                      const uint16_t* __restrict__ slab_all, int64_t n, int K,
                                                      uint32_t* __restrict__ partial_all) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint16_t* slab = slab_all + (size_t)blockIdx.y * n * 16;
    uint32_t* dst = partial_all + (size_t)blockIdx.y * ((size_t)K * K * 16);
    const uint4* ra = reinterpret_cast<const uint4*>(slab + (size_t)erow[e] * 16);
    const uint4* rb = reinterpret_cast<const uint4*>(slab + (size_t)indices[e] * 16);
    const uint4 a0 = ra[0], a1 = ra[1], b0 = rb[0], b1 = rb[1];
    const uint32_t wa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const uint32_t wb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint32_t la = (wa[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu, lb = (wb[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu;
        atomicAdd(&dst[((size_t)la * K + lb) * 16 + b], 1u);
    }
}

// injected permutations: lab[(p)*n + i] (perm-major, host order) -> slab rows
template <int B>
__global__ __launch_bounds__(256) void k_transpose_labels(int64_t n, const uint8_t* __restrict__ lab, int64_t nperm_valid,
                                                          uint8_t* __restrict__ slab_all, int pw) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int batch = blockIdx.y;
    uint32_t w[B / 4];
#pragma unroll
    for (int k = 0; k < B / 4; ++k) w[k] = 0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        int64_t p = (int64_t)batch * B + b;
        w[b >> 2] |= (p < nperm_valid ? (uint32_t)lab[(size_t)p * n + i] : 0u) << (8 * (b & 3));
    }
    if constexpr (B == 16) {
        slab_store16(slab_all + (size_t)batch * n * 16, n, i, pw, w[0], w[1], w[2], w[3]);
    } else {
        uint4* dst = reinterpret_cast<uint4*>(slab_all + ((size_t)batch * n + i) * B);
#pragma unroll
        for (int v = 0; v < B / 16; ++v) dst[v] = make_uint4(w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]);
    }
}

// ---------------------------------------------------------------------------------------------- counting
// Block -> edge chunk with XCD affinity: hardware places block b on XCD (b % 8); give each XCD one
// contiguous eighth of the edge list so halo rows shared by neighbouring chunks stay in one L2.
__device__ __forceinline__ int xcd_chunk(int b, int nblk) {
    if (nblk % 8 != 0) return b;
    return (b & 7) * (nblk >> 3) + (b >> 3);
}

// B = 16 | 32 permutations; 4 lanes per edge, lane q owns the B/4 permutations [q*B/4, (q+1)*B/4) = B/4 label
// bytes of the slab rows of both endpoints.  LDS histogram layout: word = pair*B + b  (pair = la*K + lb).
// Step s of lane (edge slot el, q) handles byte (s + el) mod B/4, so the 32 lanes of a DS lane group touch
// B distinct banks (conflict-free for B = 32).  The stagger costs nothing per step: the label words are rotated
// once per edge (v_alignbit) so that byte extraction is static, and the per-step bank offsets are loop invariant.
// pair = byte<BYTE>(la) * K + byte<BYTE>(lb) in two instructions: gfx9 sub-dword addressing folds the byte extraction
// into the 24-bit multiply and into the add (the compiler emits v_bfe_u32 x2 + v_mad_u32_u24 otherwise).
#define SQGR_PAIR_SDWA(BYTE)                                                                                              \
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #BYTE " src1_sel:DWORD"       \
        : "=v"(t)                                                                                                         \
        : "v"(la), "v"(K));                                                                                               \
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #BYTE " src1_sel:DWORD"           \
        : "=v"(p)                                                                                                         \
        : "v"(lb), "v"(t));
template <int BYTE>
__device__ __forceinline__ uint32_t pair_index(uint32_t la, uint32_t lb, uint32_t K) {
    uint32_t t, p;
    if constexpr (BYTE == 0) { SQGR_PAIR_SDWA(0) }
    else if constexpr (BYTE == 1) { SQGR_PAIR_SDWA(1) }
    else if constexpr (BYTE == 2) { SQGR_PAIR_SDWA(2) }
    else { SQGR_PAIR_SDWA(3) }
    return p;
}
#undef SQGR_PAIR_SDWA

// The edge list is either the full COO view or, on structurally symmetric graphs, the half list of sqgr_graph (edges
// r < c, then the self loops from `self_begin` on): k_reduce then forms count = h + h^T.  SELF (the half list has self
// loops): half edges add 2, self loops 1 and k_reduce halves the sum — exact, every sum is even by construction.
// List entries hold BYTE OFFSETS of the endpoints' 16-byte slab rows (16*r, 16*c) and the list is followed by
// LIST_PAD zero entries (sqgr_ctx.hip), so the look-ahead loads below never need a clamp.
// DOT2 (B = 16): the counter address in TWO instructions per (edge, permutation): v_perm_b32 with a PER-LANE selector picks the
// staggered label byte of both rows into the 16-bit halves (la << 16 | lb) — the selector carries the lane's rotation, so no
// v_alignbit — and v_dot2_u32_u16 forms la * (K << 6) + lb * 64 + bank offset in one VOP3P.
// CM (round 6; 16 permutations per pass, DOT2 addressing): how the block keeps its counters.  0: one 32-bit word per (pair,
// permutation), K*K pairs — K <= 50.  1 | 2: 16-BIT counters, two permutations per `ds_add_u32` word (the lane's increment is 1 or
// 1 << 16: which half its permutation owns) — twice the pairs per KB of LDS; the host cuts the edge list so that no block can
// carry a counter past 65 535 (at most 65 535 / weight edges per block).  2 additionally keeps ONE cell per UNORDERED label pair
// (the half list of a structurally symmetric graph: count = h + h^T needs {la, lb} only) at cell = hi * (hi + 1) / 2 + lo —
// K (K + 1) / 2 cells: 16 permutations fit up to K = 100 (mode 1, directed graphs: K = 71).  The sorted pair costs two
// instructions on the packed labels t = la << 16 | lb: min(t, t rotated by 16) = lo << 16 | hi as 32-bit integers; the quadratic
// term rides in the multiplier of the dot product: v_mad_u32_u16 forms {hi: bytes per cell, lo: (hi + 1) * bytes per cell / 2}
// and v_dot2_u32_u16 multiplies it with {hi: lo, lo: hi} — five instructions per counter address instead of two.
// The block's partial is then the LDS image itself (16-bit counters, no h + h^T pass): k_reduce16 expands it.
template <int B, int MIN_WAVES, bool SELF, bool DOT2 = false, int DBG = 0, int CM = 0>  // DBG (developer probes, bit mask): 1 no atomics, 2 no row gathers, 4 no list loads
__global__ __launch_bounds__(COUNT_THREADS, MIN_WAVES) void k_count(uint32_t nnz, const int2* __restrict__ coo,
                                                                    const uint8_t* __restrict__ slab_all, int64_t n, int K,
                                                                    int hist_words, uint32_t edges_per_block,
                                                                    uint32_t self_begin, int add_transposed,
                                                                    uint32_t* __restrict__ partial_all) {
    extern __shared__ uint32_t hist[];
    // this kernel is bound by LDS-atomic and VALU issue; when the VALU-bound shuffle kernel of the next launch group
    // shares the CU (SQGR_NHOOD_STREAMS=2), issue priority keeps the LDS pipe fed
    __builtin_amdgcn_s_setprio(3);
    constexpr int BPL = B / 4;  // label bytes per lane
    constexpr int LOGW = CM ? 5 : ((B == 32) ? 7 : 6);  // log2(bytes of one pair's B counters)
    static_assert(CM == 0 || (DOT2 && B == 16 && DBG == 0), "16-bit counters: the dot2 path at 16 permutations per pass");
    const int tid = threadIdx.x;
    for (int i = tid; i < hist_words; i += COUNT_THREADS) hist[i] = 0;
    __syncthreads();

    const uint8_t* slab = slab_all + (size_t)blockIdx.y * n * B;
    uint32_t chunk = (uint32_t)xcd_chunk(blockIdx.x, gridDim.x);
    // add_transposed >= 2 (round 6, !SELF): the SPLIT list of a directed graph (sqgr_graph::ensure_split) — `self_begin` mutual pairs
    // walked by the first add_transposed >> 2 chunks (their partial is h + h^T), LIST_PAD zeros, then the nnz - self_begin edges
    // without a mirror walked by the other chunks (their partial is h)
    bool transposed = add_transposed == 1;
    uint32_t lim = nnz;
    if (add_transposed >= 2) {
        const uint32_t nblk1 = (uint32_t)add_transposed >> 2;
        transposed = chunk < nblk1;
        if (transposed) {
            lim = self_begin;
        } else {
            chunk -= nblk1;
            coo += self_begin + (uint32_t)LIST_PAD;
            lim = nnz - self_begin;
        }
    }
    const uint32_t e0 = chunk * edges_per_block;
    const uint32_t e1 = min(lim, e0 + edges_per_block);
    // block-uniform: every edge slot of every iteration is a real edge of one weight -> no per-edge bookkeeping at all
    const bool uniform_block = (e0 + edges_per_block <= lim) && (!SELF || e0 + edges_per_block <= self_begin);
    const uint32_t q = tid & 3;
    const uint32_t el = tid >> 2;
    const uint32_t rot = (el & (BPL - 1)) * 8;  // bits to rotate right
    uint32_t bank_ofs[BPL];                      // byte offset of this lane's counter inside a pair, per step
#pragma unroll
    for (int s = 0; s < BPL; ++s) {
        const uint32_t byte = (s + el) & (BPL - 1);
        bank_ofs[s] = CM ? (q * (BPL / 2) + (byte >> 1)) * 4 : (q * BPL + byte) * 4;  // CM: permutations 2w, 2w + 1 share word w
    }
    // CM: the half of the word the lane's permutation owns alternates with the step ((s + el) & 1): two shift amounts per lane
    const uint32_t half_sh0 = CM ? 16u * (el & 1u) : 0u, half_sh1 = CM ? 16u * ((el + 1u) & 1u) : 0u;
    const uint32_t qoff = q * BPL;
    char* hist_bytes = reinterpret_cast<char*>(hist);
    static_assert(!DOT2 || B == 16, "the dot2 address path is written for 16 permutations per pass");
    uint32_t sel[BPL];  // DOT2: byte (s + el) & 3 of the b row -> bits 0..7, of the a row -> bits 16..23, zeros elsewhere
#pragma unroll
    for (int s = 0; s < BPL; ++s) sel[s] = 0x0c000c00u | ((4u + ((s + el) & 3u)) << 16) | ((s + el) & 3u);
    const uint32_t dot_k = ((uint32_t)K << (LOGW + 16)) | (1u << LOGW);  // {hi: K << 6, lo: 64}
    constexpr uint32_t tri_k = (1u << (LOGW + 16)) | (1u << (LOGW - 1));   // CM 2: {hi: bytes per cell, lo: half of them} (+ hi * half)
    uint32_t dbg_acc = 0;
    const uint32_t dbg_lin = (uint32_t)((((size_t)blockIdx.x * 40503u) % (size_t)(n - 4096)) * 16) + tid * 4;
    if constexpr (DOT2) {
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)hist;
#pragma unroll
        for (int s = 0; s < BPL; ++s) bank_ofs[s] += lds_base;
    }

    // A quad of lanes shares U = 4 consecutive edges per iteration; lane q owns the B/4 permutations [q*B/4, (q+1)*B/4) of
    // each.  Lane q loads the offset pair of edge eb + q with ONE 8-byte load (a wave reads 64 consecutive pairs: one
    // coalesced 512-byte request) and the quad exchanges them with DPP quad_perm broadcasts (VALU, no LDS, no extra memory
    // instructions: letting every lane load all four pairs itself quadruples the bytes through the vector memory pipe
    // and was measured 10 % slower).  All slab-row gathers of the 4 edges are then in flight together, so the two
    // dependent memory latencies are paid once per 4 edges.
    // Software pipeline, two stages deep: while iteration k is histogrammed, the slab rows of iteration k+1 and the
    // offset pairs of iteration k+2 are in flight (every load gets a full processing phase of slack).
    constexpr int U = 4;
    constexpr uint32_t STEP = (COUNT_THREADS / 4) * U;
    static_assert(7 * STEP + U <= LIST_PAD, "look-ahead exceeds the padding of the edge lists");
    using Row = typename std::conditional<B == 16, uint32_t, uint2>::type;
    // offsets of the 4 edges' label rows for THIS lane: the quad broadcast rides on the add of the lane's byte offset inside
    // the row (v_add_u32 with a DPP quad_perm source: one instruction instead of a broadcast and an add)
#define SQGR_ADD_DPP(dst, src, sel)                                                                      \
    asm("v_add_u32_dpp %0, %1, %2 quad_perm:[" sel "] row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src), "v"(qoff))
    struct Pairs { uint32_t r[U], c[U]; };
    auto load_pair = [&](uint32_t e) {
        if constexpr ((DBG & 4) != 0) {  // constant offsets: no list traffic, no extra VALU
            int2 v = make_int2((int)(q * 64u), (int)(q * 64u + 16u));
            asm volatile("" : "+v"(v.x), "+v"(v.y));
            return v;
        } else {
            return coo[e + q];
        }
    };
    auto spread = [&](const int2 mine) {
        Pairs pr;
        if constexpr (B == 16) {
            SQGR_ADD_DPP(pr.r[0], mine.x, "0,0,0,0"); SQGR_ADD_DPP(pr.c[0], mine.y, "0,0,0,0");
            SQGR_ADD_DPP(pr.r[1], mine.x, "1,1,1,1"); SQGR_ADD_DPP(pr.c[1], mine.y, "1,1,1,1");
            SQGR_ADD_DPP(pr.r[2], mine.x, "2,2,2,2"); SQGR_ADD_DPP(pr.c[2], mine.y, "2,2,2,2");
            SQGR_ADD_DPP(pr.r[3], mine.x, "3,3,3,3"); SQGR_ADD_DPP(pr.c[3], mine.y, "3,3,3,3");
        } else {  // 32-byte rows: list offsets are for 16-byte rows
            int2 m2 = make_int2(mine.x * 2, mine.y * 2);
            asm volatile("s_nop 1" : "+v"(m2.x), "+v"(m2.y));  // a VALU result read through DPP needs two wait states
            SQGR_ADD_DPP(pr.r[0], m2.x, "0,0,0,0"); SQGR_ADD_DPP(pr.c[0], m2.y, "0,0,0,0");
            SQGR_ADD_DPP(pr.r[1], m2.x, "1,1,1,1"); SQGR_ADD_DPP(pr.c[1], m2.y, "1,1,1,1");
            SQGR_ADD_DPP(pr.r[2], m2.x, "2,2,2,2"); SQGR_ADD_DPP(pr.c[2], m2.y, "2,2,2,2");
            SQGR_ADD_DPP(pr.r[3], m2.x, "3,3,3,3"); SQGR_ADD_DPP(pr.c[3], m2.y, "3,3,3,3");
        }
        return pr;
    };
#undef SQGR_ADD_DPP
    auto gather_rows = [&](const Pairs& pr, Row (&ra)[U], Row (&rb)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr ((DBG & 2) != 0 && B == 16) {  // real (random) labels through COALESCED loads: the LDS pattern without the gathers
                ra[u] = *reinterpret_cast<const Row*>(slab + dbg_lin + u * 8192);
                rb[u] = *reinterpret_cast<const Row*>(slab + dbg_lin + u * 8192 + 4096);
            } else {
                ra[u] = *reinterpret_cast<const Row*>(slab + pr.r[u]);
                rb[u] = *reinterpret_cast<const Row*>(slab + pr.c[u]);
            }
        }
    };
    auto histogram = [&](const Row (&row_a)[U], const Row (&row_b)[U], uint32_t eb, auto general_tag) {
        constexpr bool GENERAL = decltype(general_tag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t inc = SELF ? 2u : 1u;
            if constexpr (GENERAL) {  // branch-free tail: out-of-range edges add 0, self loops 1
                inc = (eb + u < e1) ? 1u : 0u;
                if constexpr (SELF) inc += (eb + u < min(e1, self_begin)) ? 1u : 0u;
            }
            uint32_t la[2], lb[2];
            if constexpr (DOT2) {
                la[0] = row_a[u];
                lb[0] = row_b[u];
            } else if constexpr (B == 16) {
                la[0] = __builtin_amdgcn_alignbit(row_a[u], row_a[u], rot);
                lb[0] = __builtin_amdgcn_alignbit(row_b[u], row_b[u], rot);
            } else {
                const bool sw = (rot & 32) != 0;
                const uint32_t alo = sw ? row_a[u].y : row_a[u].x, ahi = sw ? row_a[u].x : row_a[u].y;
                const uint32_t blo = sw ? row_b[u].y : row_b[u].x, bhi = sw ? row_b[u].x : row_b[u].y;
                la[0] = __builtin_amdgcn_alignbit(ahi, alo, rot);
                la[1] = __builtin_amdgcn_alignbit(alo, ahi, rot);
                lb[0] = __builtin_amdgcn_alignbit(bhi, blo, rot);
                lb[1] = __builtin_amdgcn_alignbit(blo, bhi, rot);
            }
            if constexpr (DOT2) {  // 4 addresses in 4 registers, then the 4 atomics back to back
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                uint32_t addr[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t t = __builtin_amdgcn_perm(la[0], lb[0], sel[s]);  // la << 16 | lb
                    if constexpr (CM == 2) {
                        const uint32_t tr = __builtin_amdgcn_alignbit(t, t, 16);     // lb << 16 | la
                        const uint32_t m = t < tr ? t : tr;                          // lo << 16 | hi
                        uint32_t mul;                                                // {hi: bytes per cell, lo: (hi + 1) * bytes per cell / 2}
                        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(mul) : "v"(m), "v"(1u << (LOGW - 1)), "v"(tri_k));
                        addr[s] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, mul), bank_ofs[s], false);
                    } else {
                        addr[s] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, t), __builtin_bit_cast(u16x2, dot_k), bank_ofs[s], false);
                    }
                }
                // (the LDS offset of the histogram is folded into bank_ofs: no per-atomic base add)
                if constexpr ((DBG & 1) != 0) {  // keep the address arithmetic alive, drop the atomics
                    asm volatile("" : : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]) : "memory");
                    continue;
                }
                if constexpr (CM != 0) {
                    const uint32_t inc0 = inc << half_sh0, inc1 = inc << half_sh1;  // (loop-invariant in whole blocks: `inc` is a constant there)
                    asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %5\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %5"
                                 :
                                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(inc0), "v"(inc1)
                                 : "memory");
                    continue;
                }
                asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %4\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %4"
                             :
                             : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(inc)
                             : "memory");
                continue;
            }
            auto bump = [&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                const uint32_t pair = pair_index<(s & 3)>(la[s >> 2], lb[s >> 2], (uint32_t)K);
                atomicAdd(reinterpret_cast<uint32_t*>(hist_bytes + ((pair << LOGW) + bank_ofs[s])), inc);
            };
            bump(std::integral_constant<int, 0>{});
            bump(std::integral_constant<int, 1>{});
            bump(std::integral_constant<int, 2>{});
            bump(std::integral_constant<int, 3>{});
            if constexpr (BPL == 8) {
                bump(std::integral_constant<int, 4>{});
                bump(std::integral_constant<int, 5>{});
                bump(std::integral_constant<int, 6>{});
                bump(std::integral_constant<int, 7>{});
            }
        }
    };
    auto sweep = [&](auto general_tag) {
        // (quad el owns the 4 consecutive list entries 4 * el .. 4 * el + 3: the four gather instructions of a stage then walk the
        // SAME ~21 spots' rows.  Giving one instruction 16 consecutive entries instead — a quarter of the lines per instruction —
        // measured 24 % SLOWER: the repeated touches of a line are what the L1 serves best.)
        uint32_t e = e0 + el * U;
        if constexpr (DOT2) {
            // Three stages deep: while iteration t-2 is histogrammed, the slab rows of iterations t-1 and t and the offset
            // pairs of iterations t+1 and t+2 are in flight (PMC of the two-stage loop: waves parked on vmcnt for 47 % of
            // their cycles while the LDS pipe idled a third of the time).  vmcnt retires in order and the compiler derives
            // ONE wait per loop position from all paths into it, so (i) the offset pair a stage spreads was issued BEFORE the
            // rows still in flight (pairs run two stages ahead of their gather), (ii) there is no prologue/epilogue code with
            // a different issue pattern: the loop runs T + 2 stages, the first two histogram nothing, and two dummy row
            // gathers in front of it put the same 18 loads in flight that a steady-state stage sees.
            Row ra[3][U], rb[3][U];
            int2 pr[3];
            const uint32_t T = (e1 - e0 + STEP - 1) / STEP;  // block-uniform: chunks are whole iterations
            Pairs dummy;
#pragma unroll
            for (int u = 0; u < U; ++u) dummy.r[u] = dummy.c[u] = qoff;
            pr[0] = load_pair(e);
            gather_rows(dummy, ra[1], rb[1]);
            pr[1] = load_pair(e + STEP);
            gather_rows(dummy, ra[2], rb[2]);
            for (uint32_t j = 0; j < T + 2; j += 3) {
#pragma unroll
                for (int st = 0; st < 3; ++st) {  // stage t = j + st: buffers t % 3 = st
                    pr[(st + 2) % 3] = load_pair(e + (st + 2) * STEP);       // pairs of iteration t + 2
                    gather_rows(spread(pr[st]), ra[st], rb[st]);               // rows of iteration t
                    if (j + st >= 2 && j + st - 2 < T)
                        histogram(ra[(st + 1) % 3], rb[(st + 1) % 3], e + st * STEP - 2 * STEP, general_tag);  // iteration t - 2
                }
                e += 3 * STEP;
            }
            return;
        }
        Row p_a[U], p_b[U], q_a[U], q_b[U];  // ping-pong row buffers: no register rotation at the end of an iteration
        gather_rows(spread(load_pair(e)), p_a, p_b);                      // rows of iteration 0
        int2 nxt = load_pair(e + STEP);                                    // pairs of iteration 1 (list padding: in bounds)
        while (e < e1) {
            gather_rows(spread(nxt), q_a, q_b);                            // rows of iteration k+1
            nxt = load_pair(e + 2 * STEP);                                 // pairs of iteration k+2
            histogram(p_a, p_b, e, general_tag);
            e += STEP;
            if (e >= e1) break;
            gather_rows(spread(nxt), p_a, p_b);
            nxt = load_pair(e + 2 * STEP);
            histogram(q_a, q_b, e, general_tag);
            e += STEP;
        }
    };
    if (e0 < lim) {  // (chunks past the end of a short list stay empty; their look-ahead would leave the padding)
        if (uniform_block)
            sweep(std::false_type{});
        else
            sweep(std::true_type{});
    }
    if constexpr ((DBG & 1) != 0) hist[tid] = dbg_acc;
    // The ds_add_u32 of the DOT2 path are inline asm: the compiler's wait-count pass does not know them, and the release fence of
    // __syncthreads() is a "soft" wait it drops when it sees no pending LDS operation — the barrier was reached with atomics still
    // in flight (found in round 5 on the pass kernel below: one increment in ~5e5 cells lost now and then; this kernel had the
    // same barrier without a wait in its ISA, never caught by a test).  The wait is explicit.
#ifndef SQGR_DEBUG_NO_FLUSH_WAIT  // (tools/soak_negative.sh builds without it to show that tests/test_soak_gpu.py catches the race)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    uint32_t* dst = partial_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * hist_words;
    if constexpr (CM != 0) {  // the partial is the LDS image: 16-bit counters [cell][16], expanded (and h + h^T resolved) by k_reduce16
        for (int i = tid; i < hist_words; i += COUNT_THREADS) dst[i] = hist[i];
        return;
    }
    if (transposed) {  // half list: the block's contribution to count = h + h^T is formed here, out of LDS
        constexpr int LOGB = (B == 32) ? 5 : 4;
        for (int i = tid; i < hist_words; i += COUNT_THREADS) {
            const int pair = i >> LOGB, la = pair / K, lb = pair - la * K;
            dst[i] = hist[i] + hist[((lb * K + la) << LOGB) + (i & (B - 1))];
        }
    } else {
        for (int i = tid; i < hist_words; i += COUNT_THREADS) dst[i] = hist[i];
    }
}

// 51 <= K <= 202 (round 5): K*K*16 counters no longer fit the 160 KB of LDS, so a block counts only B = 8 | 4 | 2 | 1 of the slab's 16
// permutations — one PASS — with k_count's machinery: the (half) edge list, 4-byte gathers out of the 16-byte label rows, the
// three-stage pipeline without prologue code, the counter address in two instructions (v_perm_b32 + v_dot2_u32_u16), h + h^T
// out of LDS.  What changes with B is how many lanes share an edge: LPE = 2 lanes at B = 8, each owning ONE dword of the row = 4
// permutations (NS = 4 atomics per edge and lane); one lane per edge from B = 4 down, using NS = 4 | 2 | 1 bytes of its dword.
// Every lane handles U = 4 edges per iteration whatever LPE is, so a wavefront covers J = 64 / LPE edges per gather instruction,
// 4 * J per iteration, and loads its list entries in 16- and 32-byte pieces per lane (entries 4j .. 4j + 3 of slot j).
// WHICH edges a gather instruction covers decides what the kernel costs (PMC, round 5: with entry 4j + u in instruction u — every
// fourth edge of a 128- or 256-edge run — the 32 or 64 label rows of one instruction spread over 4x and 8x the cache lines of
// k_count's 16 rows and the L1 access rate, k_count's own limiter, capped the kernel at 3.2x k_count's time per permutation).
// The list is therefore read in a permuted order (sqgr_graph::pass_list): instruction u of slot j gets entry
// (u/R)*(J*R) + j*R + u%R of its group, R = 1: J consecutive edges per instruction.
// The 16 / B passes of one edge chunk are separate blocks, CONSECUTIVE in the dispatch order of ONE XCD (block id -> XCD id % 8):
// they walk the same chunk at the same time, so it crosses the fabric once per chunk and batch and the other passes find it in
// that XCD's L2 (PMC: L2 hit rate 0.70 / 0.86 / 0.97 at 2 / 4 / 16 passes, fabric reads 1.4x k_count's whatever the number of
// passes).  (The round-1 fallback this replaces, k_count_wide<8/4/2/1>, walked the FULL CSR edge by edge with byte loads, one
// 16 / B-launch sequence per batch and no pipeline.)
// SPLIT = 2 (203 <= K <= 256: not even one permutation's K*K counters fit): a block keeps the rows la of one HALF of the labels
// only and skips the other edges' atomics — twice the blocks, every one of them walking the whole chunk; h + h^T is then formed
// by k_reduce (sym bit 2), a block does not hold the transposed rows.
// PACK: the list in 4 bytes per entry (sqgr_graph::packed_list: (col - row) << 8 | row - base of the wavefront's group; spot indices,
// scaled to plane offsets here) — half the list bytes through L2 -> L1, which is what bounds this kernel.
// CM: the counter modes of k_count — 1: 16-bit counters on K*K pairs, 2: 16-bit counters on the K (K + 1) / 2 unordered pairs of a
// symmetric graph's half list — at 8 | 4 | 2 permutations per pass: 8 up to K = 142 (mode 1: 101), 4 up to 201 (143), 2 up to 285
// (202); the partial of a block is its LDS image (`cells` cells of B 16-bit counters), written as one contiguous run.
template <int LPE, int NS, bool SELF, int SPLIT = 1, bool PACK = false, int CM = 0>
__global__ __launch_bounds__(COUNT_THREADS, 4) void k_count_pass(uint32_t nnz, const int2* __restrict__ coo, const uint32_t* __restrict__ gbase,
                                                                 const uint8_t* __restrict__ slab_all, int64_t n, int K,
                                                                 uint32_t edges_per_chunk, uint32_t self_begin, int add_transposed,
                                                                 int nchunks, uint32_t R, uint32_t* __restrict__ partial_all, int cells) {
    extern __shared__ uint32_t hist[];
    static_assert((LPE == 2 || LPE == 1) && (NS == 4 || NS == 2 || NS == 1) && (NS == 4 || LPE == 1) && (SPLIT == 1 || NS == 1), "shape");
    static_assert(CM == 0 || (NS >= 2 && SPLIT == 1), "16-bit counters: two permutations per word, whole histograms");
    constexpr int B = NS == 4 ? LPE * 4 : NS;  // permutations per pass
    constexpr int P = (16 / B) * SPLIT;        // blocks per chunk and batch of 16: passes x row halves
    constexpr int U = 4;
    constexpr uint32_t J = 64 / LPE;           // edge slots of a wavefront
    constexpr uint32_t STEP = (COUNT_THREADS / LPE) * U;
    static_assert(7 * STEP + 2 * U <= LIST_PAD, "look-ahead exceeds the padding of the edge lists");
    const int tid = threadIdx.x;
    // block -> (chunk, pass[, half]): XCD x takes the x-th eighth of the chunks and runs the P blocks of a chunk back to back
    int chunk, pass;
    if ((nchunks & 7) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        pass = j % P;
        chunk = x * (nchunks >> 3) + j / P;
    } else {
        pass = blockIdx.x % P;
        chunk = blockIdx.x / P;
    }
    const int half = pass % SPLIT;
    pass /= SPLIT;
    const int Kh = (K + SPLIT - 1) / SPLIT, a0 = half * Kh;          // this block's rows of the histogram: la in [a0, a0 + rows)
    const int rows = min(Kh, K - a0);
    const int hist_words = CM ? cells * (B / 2) : rows * K * B;
    for (int i = tid; i < hist_words; i += COUNT_THREADS) hist[i] = 0;
    __syncthreads();
    const uint8_t* slab = slab_all + (size_t)blockIdx.y * n * 16 + (size_t)pass * n * B;  // the pass's plane [n][B] (slab_store16)
    constexpr uint32_t base_byte = 0;
    const uint32_t e0 = (uint32_t)chunk * edges_per_chunk;
    const uint32_t e1 = min(nnz, e0 + edges_per_chunk);
    const bool uniform_block = (e0 + edges_per_chunk <= nnz) && (!SELF || e0 + edges_per_chunk <= self_begin);
    const uint32_t q = tid & 3;
    const uint32_t d = LPE == 2 ? (q & 1u) : 0u;   // the lane's dword inside the pass
    const uint32_t qoff = d * 4;
    const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
    const uint32_t slot = LPE == 2 ? 2 * (lane >> 2) + (q >> 1) : lane;  // edge slot j of this lane inside its wavefront
    const uint32_t el = LPE == 2 ? (uint32_t)tid >> 1 : (uint32_t)tid;   // stagger of the byte order
    uint32_t lane_log[U];  // where the lane's U edges lie in the list's own order (tails only), relative to the iteration's first edge
#pragma unroll
    for (int u = 0; u < U; ++u) lane_log[u] = wave * (4 * J) + ((uint32_t)u / R) * (J * R) + slot * R + ((uint32_t)u % R);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)hist;
    uint32_t bank_ofs[NS], sel[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t byte = (s + el) & (NS - 1);
        bank_ofs[s] = CM ? ((d * NS + byte) >> 1) * 4 + lds_base                       // permutations 2w, 2w + 1 of the pass share word w
                         : (d * NS + byte) * 4 + lds_base - (uint32_t)(a0 * K * B * 4);  // (wraps; row a0 lands on the first counter)
        sel[s] = 0x0c000c00u | ((4u + base_byte + byte) << 16) | (base_byte + byte);  // b row byte -> bits 0..7, a row byte -> bits 16..23
    }
    constexpr uint32_t CB = CM ? B * 2 : B * 4;  // bytes of one pair's (cell's) B counters
    const uint32_t dot_k = ((uint32_t)(K * CB) << 16) | CB;  // {hi: bytes per la row, lo: bytes per pair}
    constexpr uint32_t tri_k = (CB << 16) | (CB / 2);         // CM 2: {hi: bytes per cell, lo: half of them} (+ hi * half: see k_count)
    // CM: which half of its word the lane's permutation of step s owns alternates with s ((s + el) & 1)
    const uint32_t half_sh0 = CM ? 16u * (el & 1u) : 0u, half_sh1 = CM ? 16u * ((el + 1u) & 1u) : 0u;

    struct Pairs { uint32_t r[U], c[U]; };
    // what a lane loads per iteration: its U / LPE... entries — LPE 2: entries 2q, 2q + 1 of the quad's eight; LPE 1: its own four —
    // as (row, col) offset pairs (8 bytes each) or PACKed words (4 bytes each) + the base spot of the wavefront's group
    constexpr int NE = LPE == 1 ? 4 : 2;
    struct Loaded { uint32_t w[PACK ? NE : 2 * NE]; uint32_t base; };
    constexpr int LW = B == 8 ? 3 : (B == 4 ? 2 : (B == 2 ? 1 : 0));  // log2 of the plane's row width (PACK: spot -> byte offset)
    auto load_pair = [&](uint32_t e) {            // e: first edge of the block's iteration; physical entries 4 * slot .. + 3 of the wavefront's group
        Loaded L;
        L.base = 0;
        if constexpr (PACK) {
            const uint32_t* lst = reinterpret_cast<const uint32_t*>(coo) + e + wave * (4 * J) + lane * NE;
            L.base = gbase[(e >> (LPE == 2 ? 7 : 8)) + wave];  // one word per group of 4*J entries: the same address in all 64 lanes
            if constexpr (LPE == 2) {
                const uint2 v = *reinterpret_cast<const uint2*>(lst);
                L.w[0] = v.x; L.w[1] = v.y;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(lst);
                L.w[0] = v.x; L.w[1] = v.y; L.w[2] = v.z; L.w[3] = v.w;
            }
        } else {
            const int2* grp = coo + e + wave * (4 * J) + lane * NE;
            const uint4 v = *reinterpret_cast<const uint4*>(grp);
            L.w[0] = v.x; L.w[1] = v.y; L.w[2] = v.z; L.w[3] = v.w;
            if constexpr (LPE == 1) {
                const uint4 v2 = *reinterpret_cast<const uint4*>(grp + 2);
                L.w[4] = v2.x; L.w[5] = v2.y; L.w[6] = v2.z; L.w[7] = v2.w;
            }
        }
        return L;
    };
    // quad_perm [0,0,2,2] = 0xA0, [1,1,3,3] = 0xF5: lanes {0, 1} of a quad handle its entries 0..3 (slot 2t), lanes {2, 3} entries 4..7
#define SQGR_DPP(src, ctrl) (uint32_t)__builtin_amdgcn_mov_dpp((int)(src), ctrl, 0xf, 0xf, true)
    auto unpack = [&](uint32_t word, uint32_t base, uint32_t& r_off, uint32_t& c_off) {
        r_off = ((base + (word & 255u)) << LW) + qoff;
        c_off = r_off + (uint32_t)(((int32_t)word >> 8) << LW);  // (two's complement: the shift of a negative difference is the intended product)
    };
    auto spread = [&](const Loaded& L) {
        Pairs pr;
        if constexpr (PACK) {
            uint32_t e4[U];
            if constexpr (LPE == 2) {
                e4[0] = SQGR_DPP(L.w[0], 0xA0); e4[1] = SQGR_DPP(L.w[1], 0xA0);
                e4[2] = SQGR_DPP(L.w[0], 0xF5); e4[3] = SQGR_DPP(L.w[1], 0xF5);
            } else {
                e4[0] = L.w[0]; e4[1] = L.w[1]; e4[2] = L.w[2]; e4[3] = L.w[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) unpack(e4[u], L.base, pr.r[u], pr.c[u]);
        } else if constexpr (LPE == 2) {
            pr.r[0] = SQGR_DPP(L.w[0], 0xA0) + qoff; pr.c[0] = SQGR_DPP(L.w[1], 0xA0) + qoff;
            pr.r[1] = SQGR_DPP(L.w[2], 0xA0) + qoff; pr.c[1] = SQGR_DPP(L.w[3], 0xA0) + qoff;
            pr.r[2] = SQGR_DPP(L.w[0], 0xF5) + qoff; pr.c[2] = SQGR_DPP(L.w[1], 0xF5) + qoff;
            pr.r[3] = SQGR_DPP(L.w[2], 0xF5) + qoff; pr.c[3] = SQGR_DPP(L.w[3], 0xF5) + qoff;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pr.r[u] = L.w[2 * u];
                pr.c[u] = L.w[2 * u + 1];
            }
        }
        return pr;
    };
#undef SQGR_DPP
    auto gather_rows = [&](const Pairs& pr, uint32_t (&ra)[U], uint32_t (&rb)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {  // list entries of a pass list are byte offsets into a plane: B * spot
            if constexpr (NS == 4) {
                ra[u] = *reinterpret_cast<const uint32_t*>(slab + pr.r[u]);
                rb[u] = *reinterpret_cast<const uint32_t*>(slab + pr.c[u]);
            } else if constexpr (NS == 2) {
                ra[u] = *reinterpret_cast<const uint16_t*>(slab + pr.r[u]);
                rb[u] = *reinterpret_cast<const uint16_t*>(slab + pr.c[u]);
            } else {
                ra[u] = slab[pr.r[u]];
                rb[u] = slab[pr.c[u]];
            }
        }
    };
    auto histogram = [&](const uint32_t (&row_a)[U], const uint32_t (&row_b)[U], uint32_t eb, auto general_tag) {
        constexpr bool GENERAL = decltype(general_tag)::value;
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t inc = SELF ? 2u : 1u;
            if constexpr (GENERAL) {  // branch-free tail: out-of-range edges add 0, self loops 1
                const uint32_t idx = eb + lane_log[u];
                inc = (idx < e1) ? 1u : 0u;
                if constexpr (SELF) inc += (idx < min(e1, self_begin)) ? 1u : 0u;
            }
            uint32_t addr[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t t = __builtin_amdgcn_perm(row_a[u], row_b[u], sel[s]);  // la << 16 | lb
                if constexpr (CM == 2) {
                    const uint32_t tr = __builtin_amdgcn_alignbit(t, t, 16);
                    const uint32_t m = t < tr ? t : tr;  // lo << 16 | hi
                    uint32_t mul;
                    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(mul) : "v"(m), "v"(CB / 2), "v"(tri_k));
                    addr[s] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, mul), bank_ofs[s], false);
                } else {
                    addr[s] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, t), __builtin_bit_cast(u16x2, dot_k), bank_ofs[s], false);
                }
            }
            if constexpr (CM != 0) {
                const uint32_t inc0 = inc << half_sh0, inc1 = inc << half_sh1;
                if constexpr (NS == 4)
                    asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %5\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %5"
                                 :
                                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(inc0), "v"(inc1)
                                 : "memory");
                else
                    asm volatile("ds_add_u32 %0, %2\n\tds_add_u32 %1, %3" : : "v"(addr[0]), "v"(addr[1]), "v"(inc0), "v"(inc1) : "memory");
            } else if constexpr (NS == 4)
                asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %4\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %4"
                             :
                             : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(inc)
                             : "memory");
            else if constexpr (NS == 2)
                asm volatile("ds_add_u32 %0, %2\n\tds_add_u32 %1, %2" : : "v"(addr[0]), "v"(addr[1]), "v"(inc) : "memory");
            else if constexpr (SPLIT == 1)
                asm volatile("ds_add_u32 %0, %1" : : "v"(addr[0]), "v"(inc) : "memory");
            else if (addr[0] - lds_base < (uint32_t)hist_words * 4u)  // la in this block's half of the rows (unsigned: rows below a0 wrap)
                asm volatile("ds_add_u32 %0, %1" : : "v"(addr[0]), "v"(inc) : "memory");
        }
    };
    auto sweep = [&](auto general_tag) {
        // the uniform three-stage loop of k_count (T + 2 stages, two dummy gathers in front: one wait pattern for every stage)
        uint32_t e = e0;
        uint32_t ra[3][U], rb[3][U];
        Loaded pr[3];
        const uint32_t T = (e1 - e0 + STEP - 1) / STEP;
        Pairs dummy;
#pragma unroll
        for (int u = 0; u < U; ++u) dummy.r[u] = dummy.c[u] = qoff;
        pr[0] = load_pair(e);
        gather_rows(dummy, ra[1], rb[1]);
        pr[1] = load_pair(e + STEP);
        gather_rows(dummy, ra[2], rb[2]);
        for (uint32_t j = 0; j < T + 2; j += 3) {
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                pr[(st + 2) % 3] = load_pair(e + (st + 2) * STEP);
                gather_rows(spread(pr[st]), ra[st], rb[st]);
                if (j + st >= 2 && j + st - 2 < T)
                    histogram(ra[(st + 1) % 3], rb[(st + 1) % 3], e + st * STEP - 2 * STEP, general_tag);
            }
            e += 3 * STEP;
        }
    };
    __builtin_amdgcn_s_setprio(3);
    if (e0 < nnz) {
        if (uniform_block)
            sweep(std::false_type{});
        else
            sweep(std::true_type{});
    }
#ifndef SQGR_DEBUG_NO_FLUSH_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the inline-asm atomics are invisible to the compiler's wait counts (see k_count)
#endif
    __syncthreads();
    // the chunk's partial histogram is [16 / B planes][pair][B] (k_reduce's slot order): a block writes ONE contiguous run — its
    // plane, or its half of the plane's rows (first version: [pair][16] with the pass's B columns scattered into it — 4 bytes
    // per 64-byte line at B = 1, 7.4 GB of write traffic per 2560 permutations at K = 200 instead of 3.3)
    if constexpr (CM != 0) {  // the LDS image: [cell][B] 16-bit counters of this pass's plane (k_reduce16 expands them)
        uint32_t* dst16 = partial_all + ((size_t)blockIdx.y * nchunks + chunk) * ((size_t)cells * 8) + (size_t)pass * hist_words;
        for (int i = tid; i < hist_words; i += COUNT_THREADS) dst16[i] = hist[i];
        return;
    }
    uint32_t* dst = partial_all + ((size_t)blockIdx.y * nchunks + chunk) * ((size_t)K * K * 16) + (size_t)pass * K * K * B + (size_t)a0 * K * B;
    const float inv_k = 1.0f / (float)K;  // pair / K without an integer division: (pair + 0.5) / K is at least 0.5 / K away from an
                                          // integer, the float product is off by < 2^-22 * K (pair < 2^16, K <= 256)
    for (int i = tid; i < hist_words; i += COUNT_THREADS) {
        uint32_t v = hist[i];
        if (SPLIT == 1 && add_transposed) {
            const int pair = i / B, b = i - pair * B;
            const int la = (int)(((float)pair + 0.5f) * inv_k), lb = pair - la * K;
            v += hist[(lb * K + la) * B + b];
        }
        dst[i] = v;
    }
}

// K*K does not fit LDS even for one permutation (K > 202) -> device-scope atomics straight into the (single) partial (BE == 0).
__global__ __launch_bounds__(COUNT_THREADS) void k_count_global(int64_t nnz, const int32_t* __restrict__ erow,
                                                                const int32_t* __restrict__ indices,
                                                                const uint8_t* __restrict__ slab_all, int64_t n, int K,
                                                                int64_t edges_per_block, uint32_t* __restrict__ partial_all) {
    constexpr int B = 16;
    const int K2 = K * K;
    const int tid = threadIdx.x;
    const uint8_t* slab = slab_all + (size_t)blockIdx.y * n * B;
    uint32_t* dst = partial_all + (size_t)blockIdx.y * ((size_t)K2 * B);
    const int64_t e0 = (int64_t)blockIdx.x * edges_per_block;
    const int64_t e1 = min(nnz, e0 + edges_per_block);
    for (int64_t e = e0 + tid; e < e1; e += COUNT_THREADS) {
        const uint4 ra = *reinterpret_cast<const uint4*>(slab + (size_t)erow[e] * B);
        const uint4 rb = *reinterpret_cast<const uint4*>(slab + (size_t)indices[e] * B);
        const uint32_t wa[4] = {ra.x, ra.y, ra.z, ra.w}, wb[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const uint32_t pair = ((wa[b >> 2] >> (8 * (b & 3))) & 255u) * K + ((wb[b >> 2] >> (8 * (b & 3))) & 255u);
            atomicAdd(&dst[(size_t)pair * B + b], 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------- reduction
// Partial histograms and accumulator slots of a batch are laid out [16 / pw planes][pair][pw]: slot j <-> permutation
// b = (j / (K2 * pw)) * pw + j % pw of pair (j / pw) % K2 — pw = B (one plane, word = pair * B + b) for k_count and the
// device-scope kernels, pw = the pass width for k_count_pass, whose blocks each write ONE contiguous plane (or, for K > 202,
// one half of its rows).  acc slots are private to (batch, j): no atomics, bit-reproducible.
// 256 threads = 64 slots x 4 slices of the block loop (combined through LDS).
// sym: bit 1 (value 2): the partials are in doubled units (half lists with self loops): count = sum / 2;
//      bit 2 (value 4): the partials hold h of the half list, not h + h^T: the transposed pair's slot is added here (the
//      pass kernel with split rows cannot form it in LDS).
__global__ __launch_bounds__(256) void k_reduce(const uint32_t* __restrict__ partial_all, int nblk, int nsum, int hist_words, int B, int pw,
                                                int K, int sym, const int64_t* __restrict__ shift, int64_t perm_batch0,
                                                int64_t perm_begin, int64_t perm_end, int64_t* __restrict__ acc_sum,
                                                uint64_t* __restrict__ acc_sq, uint32_t* __restrict__ perms_out) {
    __shared__ unsigned long long part[4][64];
    const int wl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + wl;
    const int batch = blockIdx.y;
    const int K2 = K * K;
    int pair = 0, b = 0, jt = 0;
    if (j < hist_words) {
        const int plane = j / (K2 * pw), rem = j - plane * (K2 * pw);
        pair = rem / pw;
        b = plane * pw + (rem - pair * pw);
        if (sym & 4) {
            const int la = pair / K, lb = pair - la * K;
            jt = plane * (K2 * pw) + (lb * K + la) * pw + (rem - pair * pw);
        }
    }
    unsigned long long c = 0;
    if (j < hist_words) {
        const uint32_t* src = partial_all + (size_t)batch * nblk * hist_words;
        for (int k = slice; k < nsum; k += 4) {  // nblk: partials per batch (stride); nsum: how many of them hold something to add
            c += src[(size_t)k * hist_words + j];
            if (sym & 4) c += src[(size_t)k * hist_words + jt];
        }
    }
    part[slice][wl] = c;
    __syncthreads();
    if (slice != 0 || j >= hist_words) return;
    c = part[0][wl] + part[1][wl] + part[2][wl] + part[3][wl];
    if (sym & 2) c >>= 1;
    const int64_t p = perm_batch0 + (int64_t)batch * B + b;
    if (p >= perm_end || p < perm_begin) return;
    const int64_t d = (int64_t)c - shift[pair];
    const size_t slot = (size_t)batch * hist_words + j;
    acc_sum[slot] += d;
    acc_sq[slot] += (uint64_t)(d * d);
    if (perms_out) perms_out[(size_t)(p - perm_begin) * K2 + pair] = (uint32_t)c;
}

// Reduction of the 16-bit partials (counter modes 1 and 2 of k_count / k_count_pass): per (batch, chunk) the blocks wrote
// [16 / PW planes][cell][PW] uint16 — cells = K*K (mode 1) or K (K + 1) / 2 unordered pairs at cell = hi * (hi + 1) / 2 + lo (TRI).  ONE
// thread owns a cell of a batch: it sums the cell's 16 permutations over the chunks (PW / 2 words per plane and chunk: 32 contiguous
// bytes per lane at PW = 16), turns them into the counts they stand for, and adds the batch's sum of d = count - shift and of d^2
// to ONE accumulator slot per ordered pair (acc[batch][pair]: private, no atomics, bit-reproducible) — 16 times fewer accumulator
// bytes than k_reduce's slot per (pair, permutation), which at K = 200 were as many as the partials themselves.
// TRI: count[a][b] = count[b][a] = T{a,b} for a != b, count[a][a] = 2 T{a,a} (an edge inside a cluster is one half-list entry and
// two entries of the full scan); `doubled` (half list with self loops: half edges weigh 2, self loops 1): count[a][b] = T{a,b} / 2,
// count[a][a] = T{a,a} — every T{a,b}, a != b, is even by construction.
template <bool TRI, int PW>
__global__ __launch_bounds__(256) void k_reduce16(const uint32_t* __restrict__ partial_all, int nblk, int cells, int K, int doubled,
                                                  const int64_t* __restrict__ shift, int64_t perm_batch0, int64_t perm_begin, int64_t perm_end,
                                                  int64_t* __restrict__ acc_sum, uint64_t* __restrict__ acc_sq, uint32_t* __restrict__ perms_out) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int batch = blockIdx.y;
    if (cell >= cells) return;
    const int K2 = K * K;
    const size_t words = (size_t)cells * 8;  // 32-bit words of one chunk's partial
    uint32_t c[16];  // (a count is < 2^32 by the interface: uint32 per-permutation counts)
#pragma unroll
    for (int b = 0; b < 16; ++b) c[b] = 0;
    const uint32_t* src = partial_all + (size_t)batch * nblk * words + (size_t)cell * (PW / 2);
#pragma unroll 2
    for (int k = 0; k < nblk; ++k, src += words) {
#pragma unroll
        for (int plane = 0; plane < 16 / PW; ++plane) {
            const uint32_t* q = src + (size_t)plane * cells * (PW / 2);
            uint32_t w[PW / 2];
            if constexpr (PW == 16) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(q), v1 = *reinterpret_cast<const uint4*>(q + 4);
                w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
            } else if constexpr (PW == 8) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(q);
                w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
            } else if constexpr (PW == 4) {
                const uint2 v0 = *reinterpret_cast<const uint2*>(q);
                w[0] = v0.x; w[1] = v0.y;
            } else {
                w[0] = *q;
            }
#pragma unroll
            for (int i = 0; i < PW / 2; ++i) {
                c[plane * PW + 2 * i] += w[i] & 0xffffu;
                c[plane * PW + 2 * i + 1] += w[i] >> 16;
            }
        }
    }
    int pa[2], npairs = 1;
    bool diag = false;
    if constexpr (TRI) {
        int hi = (int)((sqrtf(8.0f * (float)cell + 1.0f) - 1.0f) * 0.5f);
        while (hi * (hi + 1) / 2 > cell) --hi;
        while ((hi + 1) * (hi + 2) / 2 <= cell) ++hi;
        const int lo = cell - hi * (hi + 1) / 2;
        diag = lo == hi;
        npairs = diag ? 1 : 2;
        pa[0] = lo * K + hi;
        pa[1] = hi * K + lo;
    } else {
        pa[0] = pa[1] = cell;
    }
    const int64_t p0 = perm_batch0 + (int64_t)batch * 16;
    for (int t = 0; t < npairs; ++t) {
        const int64_t sh = shift[pa[t]];
        int64_t sum = 0;
        uint64_t sq = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int64_t p = p0 + b;
            if (p >= perm_end || p < perm_begin) continue;
            uint32_t cnt = c[b];
            if constexpr (TRI) cnt = diag ? (doubled ? cnt : 2 * cnt) : (doubled ? cnt >> 1 : cnt);
            const int64_t d = (int64_t)cnt - sh;
            sum += d;
            sq += (uint64_t)(d * d);
            if (perms_out) perms_out[(size_t)(p - perm_begin) * K2 + pa[t]] = cnt;
        }
        const size_t slot = (size_t)batch * K2 + pa[t];
        acc_sum[slot] += sum;
        acc_sq[slot] += sq;
    }
}

// thread per slot j of a batch: sums the slot over the batches (coalesced whatever the plane width), then adds into its pair's
// total — 16 integer atomics per pair and moment; integer addition is exact and order-independent, so the result is
// bit-reproducible.  out_sum / out_sq are zeroed by the caller.
__global__ __launch_bounds__(256) void k_finalize(const int64_t* __restrict__ acc_sum, const uint64_t* __restrict__ acc_sq, int nbatch,
                                                  int hist_words, int pw, int K2, int64_t* __restrict__ out_sum,
                                                  uint64_t* __restrict__ out_sq) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= hist_words) return;
    int64_t s = 0;
    uint64_t q = 0;
    for (int t = 0; t < nbatch; ++t) {
        s += acc_sum[(size_t)t * hist_words + j];
        q += acc_sq[(size_t)t * hist_words + j];
    }
    const int pair = (j % (K2 * pw)) / pw;
    atomicAdd(reinterpret_cast<unsigned long long*>(out_sum) + pair, (unsigned long long)s);
    atomicAdd(reinterpret_cast<unsigned long long*>(out_sq) + pair, (unsigned long long)q);
}

// in place: chunk 0's partial of every batch becomes the sum over the batch's chunks (k_reduce with nsum = 1 then reads every
// slot — and, for sym bit 2, its transposed twin — once instead of once per chunk)
__global__ __launch_bounds__(256) void k_sum_chunks(uint32_t* __restrict__ partial_all, int nblk, int hist_words) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= hist_words) return;
    uint32_t* src = partial_all + (size_t)blockIdx.y * nblk * hist_words + j;
    uint32_t c = 0;
    for (int k = 0; k < nblk; ++k) c += src[(size_t)k * hist_words];
    src[0] = c;
}

// observed counts / interaction matrix: one pass, thread per edge
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_edge_pairs(int64_t nnz, const int32_t* __restrict__ erow,
                                                    const int32_t* __restrict__ indices, const double* __restrict__ data,
                                                    const int32_t* __restrict__ labels, int K,
                                                    unsigned long long* __restrict__ out_u64, double* __restrict__ out_f64) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int la = labels[erow[e]], lb = labels[indices[e]];
    if (la < 0 || lb < 0) return;  // masked (NaN category) spots: interaction_matrix semantics
    if (WEIGHTED)
        atomicAdd(&out_f64[la * K + lb], data[e]);  // only where K*K doubles do not fit LDS (K > 143): order-dependent rounding
    else
        atomicAdd(&out_u64[la * K + lb], 1ull);
}

// Weighted edge sums (gr/_nhood.py:412-429 sums float64 weights serially), bit-reproducible BY CONSTRUCTION: every WAVE owns
// a contiguous run of edges and a private K*K float64 accumulator in LDS, and adds its edges to it in EDGE ORDER — the 64
// edges of a trip are grouped by target cell with ballots, the lanes of a group are read out in lane (= edge) order by
// readlane and added one after the other to the cell's running value by the group's first lane.  No floating-point atomic,
// no assumption about the order in which the hardware applies conflicting LDS operations (round 2 relied on that); a wave's
// partial sums are a pure function of its inputs and k_sum_partials adds the partials in wave order.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_edge_weight_partials(int64_t nnz, const int32_t* __restrict__ erow,
                                                                      const int32_t* __restrict__ indices, const double* __restrict__ data,
                                                                      const int32_t* __restrict__ labels, int K, int64_t edges_per_wave,
                                                                      double* __restrict__ partials) {
    extern __shared__ double s_acc[];  // [WAVES][K*K]
    const int K2 = K * K, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* acc = s_acc + (size_t)wave * K2;
    for (int c = lane; c < K2; c += 64) acc[c] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int64_t w = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t e0 = w * edges_per_wave, e1 = e0 + edges_per_wave < nnz ? e0 + edges_per_wave : nnz;
    for (int64_t base = e0; base < e1; base += 64) {
        const int64_t e = base + lane;
        int cell = -1;
        double wt = 0.0;
        if (e < e1) {
            const int la = labels[erow[e]], lb = labels[indices[e]];
            if (la >= 0 && lb >= 0) {
                cell = la * K + lb;
                wt = data[e];
            }
        }
        unsigned long long todo = __ballot(cell >= 0);
        while (todo) {  // wave-uniform: one target cell per turn, first pending lane first
            const int leader = __ffsll((long long)todo) - 1;
            const int lc = __shfl(cell, leader, 64);
            unsigned long long grp = __ballot(cell == lc) & todo;
            todo &= ~grp;
            double run = (lane == leader) ? acc[lc] : 0.0;
            while (grp) {  // the group's weights in lane order = edge order
                const int l = __ffsll((long long)grp) - 1;
                run += __shfl(wt, l, 64);
                grp &= grp - 1;
            }
            if (lane == leader) acc[lc] = run;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next turn may read what this one wrote
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double* dst = partials + (size_t)w * K2;
    for (int c = lane; c < K2; c += 64) dst[c] = acc[c];
}

__global__ __launch_bounds__(64) void k_sum_partials(const double* __restrict__ partials, int64_t nparts, int K2, double* __restrict__ out) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= K2) return;
    double s = 0.0;
    for (int64_t p = 0; p < nparts; ++p) s += partials[(size_t)p * K2 + c];  // fixed order
    out[c] = s;
}

// columns of W -> slab rows of the batched count kernel: slab[(batch*n + i)*B + b] = W[pos(i)][p0 + batch*B + b]
template <int B, bool HAS_LIBS>
__global__ __launch_bounds__(256) void k_columns_to_slab(int64_t n, int64_t stride, const uint8_t* __restrict__ W, int64_t p0,
                                                         const int32_t* __restrict__ lib_of, const int32_t* __restrict__ rank_of,
                                                         const uint32_t* __restrict__ lib_off, uint8_t* __restrict__ slab_all, int pw) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int batch = blockIdx.y;
    const int64_t pos = HAS_LIBS ? (int64_t)lib_off[lib_of[i]] + rank_of[i] : i;
    const uint4* src = reinterpret_cast<const uint4*>(W + pos * stride + p0 + (int64_t)batch * B);  // 16-byte aligned
    if constexpr (B == 16) {
        const uint4 v = src[0];
        slab_store16(slab_all + (size_t)batch * n * 16, n, i, pw, v.x, v.y, v.z, v.w);
    } else {
        uint4* dst = reinterpret_cast<uint4*>(slab_all + ((size_t)batch * n + i) * B);
#pragma unroll
        for (int v = 0; v < B / 16; ++v) dst[v] = src[v];
    }
}

// rows of the numpy-stream shuffle -> slab rows of the batched count kernel, without the column matrix in between (no libraries:
// position == spot): slab[(batch*n + i)*B + b] = R[(q0 + batch*B + b) * row_stride + i].  A thread takes 4 consecutive spots:
// one dword from each of the batch's B rows (a wave reads 256 contiguous bytes per row), a 4 x B byte transposition in registers
// (v_perm_b32), 4 slab rows = 4*B contiguous bytes out.  Rows past `pc` read as label 0 (consumers read whole batches).
template <int B>
__global__ __launch_bounds__(256) void k_rows_to_slab(int64_t n, int64_t row_stride, const uint8_t* __restrict__ R, int64_t q0, int64_t pc,
                                                      uint8_t* __restrict__ slab_all, int pw) {
    const int64_t i4 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    const int batch = blockIdx.y;
    uint32_t v[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int64_t q = q0 + (int64_t)batch * B + b;
        v[b] = q < pc ? *reinterpret_cast<const uint32_t*>(R + q * row_stride + i4) : 0u;  // (row_stride is a multiple of 64 >= n)
    }
    uint8_t* dst = slab_all + ((size_t)batch * n + i4) * B;
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
        if (i4 + sp >= n) break;
        const uint32_t sel = (uint32_t)sp | ((4u + (uint32_t)sp) << 4);  // byte sp of x, byte sp of y
        uint32_t w[B / 4];
#pragma unroll
        for (int k = 0; k < B / 4; ++k) {
            const uint32_t lo = __byte_perm(v[4 * k], v[4 * k + 1], sel), hi = __byte_perm(v[4 * k + 2], v[4 * k + 3], sel);
            w[k] = __byte_perm(lo, hi, 0x5410);
        }
        if (B == 16 && pw != 16) {  // planes of the pass kernel
            slab_store16(slab_all + (size_t)batch * n * 16, n, i4 + sp, pw, w[0], w[1], w[2], w[3]);
            continue;
        }
#pragma unroll
        for (int k = 0; k < B / 16; ++k)
            reinterpret_cast<uint4*>(dst + (size_t)sp * B)[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    }
}

// numpy's float64 `perms.mean(axis=0)` and `perms.std(axis=0)` of the (P, K, K) count array, bit for bit: a reduction over
// the first axis of a C-contiguous array is a plain sequential accumulation per cell (one rounded add per permutation,
// in permutation order) — `mean = sum / P`, then `sum((x - mean)**2) / P` accumulated the same way, then sqrt.  One
// thread per cell walks the permutations in order; every operation is individually rounded (-ffp-contract=off).
// The accumulation is a CHAIN that can be cut anywhere: `acc` comes in with the sum over the permutations in front of this
// slice and leaves with this slice added — several ranks (or several segments of one rank) continue each other's running sums
// in permutation order and arrive at numpy's own sequence of additions.  mean == nullptr: acc += x; else acc += (x - mean)^2.
__global__ __launch_bounds__(64) void k_numpy_chain(const uint32_t* __restrict__ perms, int64_t count, int K2, const double* __restrict__ mean,
                                                    double* __restrict__ acc) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= K2) return;
    const uint32_t* col = perms + c;
    double a = acc[c];
    const bool second = mean != nullptr;
    const double m = second ? mean[c] : 0.0;
    // the additions are a dependent chain, the loads are not: 16 of them in flight per trip (round 5 waited one global-memory
    // latency per permutation: 5 ms of a 138 ms call at 8192 permutations); the order of the additions is unchanged
    constexpr int U = 16;
    int64_t q = 0;
    for (; q + U <= count; q += U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = col[(size_t)(q + u) * K2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (second) {
                const double d = (double)v[u] - m;
                a += d * d;
            } else {
                a += (double)v[u];
            }
        }
    }
    for (; q < count; ++q) {
        if (second) {
            const double d = (double)col[(size_t)q * K2] - m;
            a += d * d;
        } else {
            a += (double)col[(size_t)q * K2];
        }
    }
    acc[c] = a;
}
// op 0: v <- v / P (the mean from the finished sum); op 1: v <- sqrt(v / P)
__global__ __launch_bounds__(64) void k_numpy_chain_finish(double* __restrict__ v, int K2, double P, int op) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= K2) return;
    const double x = v[c] / P;
    v[c] = op ? sqrt(x) : x;
}

}  // namespace sqgr

using namespace sqgr;

// dynamic LDS above 64 KiB must be opted into per kernel
template <typename KernelT>
static int allow_lds(KernelT kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SQGR_OK;
}

struct sqgr_nhood {
    sqgr_ctx* ctx = nullptr;
    const sqgr_graph* g = nullptr;
    int64_t n = 0;
    int K = 0, K2 = 0;
    int n_libs = 1;
    bool has_libs = false;
    LibDom dom0{};
    int blk_bytes = 0;  // words of the block table (one per high digit and library)
    DevBuf<uint32_t> cum;  // [n_libs][kpad] label boundaries of the label-sorted base
    int kpad = 0;
    DevBuf<int32_t> lib_of, rank_of;
    DevBuf<LibDom> libs;
    DevBuf<uint8_t> base_pos;   // base labels in library-grouped position order (numpy-compatible mode)
    DevBuf<uint16_t> base16;    // the same as 16-bit labels (more than 256 clusters)
    DevBuf<int32_t> perm_idx;   // numpy permutations of a chunk (more than 256 clusters: labels are gathered through them)
    DevBuf<uint32_t> lib_off;   // [n_libs + 1] first position of every library
    DevBuf<uint8_t> wcol;       // [position][stride] column workspace of the numpy-compatible shuffle
    DevBuf<uint64_t> pcg_states;
    PcgWorkspace pcg_ws;        // jump-ahead table and row workspace of the numpy-compatible shuffle (sqgr_pcg.hip)
    bool has_labels = false;
    bool independent = false;   // sqgr_nhood_run under SQGR_SHUFFLE_INDEPENDENT=1: one 8-round bijection per permutation
    // tuning
    int B = 16;
    int nblk = 0;
    int nbatch = 0;  // batches per launch group; 0: automatic (resolve_tuning)
    // workspace
    DevBuf<uint32_t> keys;   // 2 buffers (ping-pong between the shuffle and the count stream)
    DevBuf<uint8_t> slab;    // 2 buffers
    hipEvent_t ev_shuffled[2] = {nullptr, nullptr}, ev_counted[2] = {nullptr, nullptr};
    ~sqgr_nhood() {
        for (int i = 0; i < 2; ++i) {
            if (ev_shuffled[i]) (void)hipEventDestroy(ev_shuffled[i]);
            if (ev_counted[i]) (void)hipEventDestroy(ev_counted[i]);
        }
    }
    size_t keys_stride() const { return (size_t)nbatch * key_words_per_row(B, n_libs); }
    bool wide() const { return K > 256; }  // 16-bit labels, device-scope counters
    bool tab_lds = true;                   // the label-boundary table fits LDS next to the block table (nhood_build)
    DevBuf<int32_t> spot_of;               // sqgr_nhood_set_spot_map: slab row i holds the labels of the caller's observation spot_of[i]
    bool mapped() const { return spot_of.p != nullptr; }
    size_t slab_stride() const { return (size_t)nbatch * n * B * (wide() ? 2 : 1); }  // bytes
    DevBuf<uint32_t> partial;
    DevBuf<int64_t> acc_sum;
    DevBuf<uint64_t> acc_sq;
    DevBuf<int64_t> shift, fin;  // fin: [K2] sum of d, then [K2] sum of d*d (uint64 bit patterns) — one all-reduce
    sqgr_comm* comm = nullptr;   // optional: moments are all-reduced on the device (sqgr_nhood_set_comm)
    DevBuf<uint32_t> perms_dev;
    DevBuf<uint8_t> stage;

    int hist_words() const { return K2 * B; }  // accumulator slots of a batch: every ordered pair x every permutation of the slab row
    // Counter mode of the LDS count kernels above 50 clusters (round 6; see k_count): 0 — 32-bit counters on K*K pairs (what
    // K <= 50 keeps, and what serves one permutation per pass: a tune / SQGR_COUNT_PASS_B cap of 1, directed graphs above 202
    // clusters); 1 — 16-bit counters on K*K pairs (directed graphs); 2 — 16-bit counters on the K (K + 1) / 2 unordered pairs
    // (structurally symmetric graphs: the half list).  Needs g->ensure_half() to have run (ensure_workspace, sqgr_nhood_info).
    int max_label_count = 0;  // largest cluster of the base labels (0: unknown — injected label vectors)
    int pass_cap = 16;
    int cap() const {
        static const int env_cap = [] { const char* e = getenv("SQGR_COUNT_PASS_B"); return e ? std::max(atoi(e), 1) : 16; }();
        return std::min(pass_cap, env_cap);
    }
    int cm() const {
        static const bool off = [] { const char* e = getenv("SQGR_COUNT_C16"); return e && atoi(e) == 0; }();
        if (off || wide() || B != 16 || !g || (size_t)K2 * 16 * 4 <= LDS_BUDGET || cap() < 2) return 0;
        const bool sym = g->sym_state == 1;
        const size_t c = sym ? (size_t)K * (K + 1) / 2 : (size_t)K2;
        if (c * 2 * 2 > LDS_BUDGET) return 0;  // not even two permutations (a directed graph above 202 clusters)
        return sym ? 2 : 1;
    }
    int cells() const { return cm() == 2 ? K * (K + 1) / 2 : K2; }
    // permutations of the 16-wide slab whose histograms fit LDS together: 16 -> k_count; 8, 4, 2, 1 -> that many per PASS of
    // k_count_pass (16 / be passes per batch); 0: not even one (32-bit counters, K > 202) -> device-scope counters (k_count_global).
    // sqgr_nhood_tune(perms_per_pass = 8|4|2|1) / SQGR_COUNT_PASS_B cap it (tests: every pass width on one input; experiments:
    // fewer permutations per pass, more blocks per CU).
    int be() const {
        if (cm()) {
            for (int b : {16, 8, 4, 2})
                if ((size_t)cells() * b * 2 <= LDS_BUDGET && b <= cap()) return b;
        }
        for (int b : {16, 8, 4, 2, 1})
            if ((size_t)K2 * b * 4 <= LDS_BUDGET && b <= cap()) return b;
        if (!wide() && (size_t)((K + 1) / 2) * K * 4 <= LDS_BUDGET) return 1;  // half of the rows per block (split() == 2)
        return 0;
    }
    int split() const { return (B == 16 && !wide() && !cm() && be() == 1 && (size_t)K2 * 4 > LDS_BUDGET) ? 2 : 1; }
    int partial_w() const { return (B == 16 && !wide() && be() > 0 && be() < 16) ? be() : B; }  // plane width of partials and accumulator slots
    bool lds_path() const { return !wide() && (B == 32 || be() > 0); }  // block-local LDS histograms, (half) edge list
    int passes() const { return (B == 16 && !wide() && be() > 0) ? (16 / be()) * split() : 1; }
    // layout of a batch's 16 * n slab bytes: 16 = rows [n][16]; 8 | 4 | 2 | 1 = 16 / w planes [n][w], one per pass (slab_store16)
    int plane_w() const { return (B == 16 && !wide() && be() > 0) ? be() : 16; }
    // 32-bit words of ONE block-partial set of a batch's chunk: K*K*B counters, or (16-bit modes) cells x 16 half words
    size_t part_words() const { return cm() ? (size_t)cells() * 8 : (size_t)hist_words(); }
    // accumulator slots of a batch and their plane width (k_finalize): one per (pair, permutation) — or, 16-bit modes, one per pair
    int acc_words() const { return cm() ? K2 : hist_words(); }
    int acc_pw() const { return cm() ? 1 : partial_w(); }
    // 16-bit counters: the most edges one block may count — no counter can pass 65 535 whatever the labels are (a block's cell
    // receives at most weight x edges; weight 2 on half lists with self loops) — in whole iterations of the kernel that runs
    uint32_t edge_step() const { return be() == 16 ? 1024u : (be() == 8 ? 2048u : 4096u); }
    int64_t list_edges() const {  // entries of the list the count kernel walks: half list | split list of a directed graph | all edges
        if (!g) return 0;
        if (lds_path() && g->sym_state == 1) return g->n_half + g->n_self;
        if (lds_path() && B == 16 && be() == 16 && !cm() && g->split_state == 1) return g->n_mutual + g->n_oneway;
        return g->nnz;
    }
    // ... unless the labels themselves bound it: a cell {a, b} receives only list entries with an endpoint in cluster a, at most
    // (size of a) x (longest row) of them, and a shuffle keeps the cluster sizes.  That argument needs every one of the slab's 16
    // columns to BE a shuffle of the base labels: `columns_valid` — true inside sqgr_nhood_run only (the device generator always
    // fills whole rows; the numpy-stream and injected-label paths leave the columns past the last permutation stale).
    bool columns_valid = false;
    int64_t chunk_cap(bool relaxed) const {
        if (!cm()) return 0;
        const uint32_t w = (g->sym_state == 1 && g->n_self > 0) ? 2u : 1u;
        if (relaxed && max_label_count > 0 && (int64_t)max_label_count * std::max<int64_t>(g->max_row_len, 1) * w <= 65535) return 0;
        return (int64_t)(65535u / w / edge_step()) * edge_step();
    }
    // 1024-thread blocks per batch of a launch with `nb` batches = edge chunks per batch (k_count_pass: times `passes()` blocks).
    // nblk > 0: fixed by sqgr_nhood_tune.  Auto: ~4 blocks per CU over the whole launch, at least 8 per batch (one per XCD), whole
    // XCD shares (a multiple of 8).  Round 2 had measured 32-48 blocks per batch as the best with 64 batches per launch; with the
    // 160-batch launches of today fewer, longer chunks win — fewer partial histograms to write and re-read (blocks * K*K*64 bytes
    // per batch), fewer ramps and tails (tools/nhood_k_sweep.py --sweep, round 5, permutations/s at 1e6 spots; blocks per batch 8 /
    // 16 / 32 / 64: K = 30 953 k / 948 k / 921 k / 864 k — round 4's 32 was 6 % behind —, K = 64 687 k / 657 k / 602 k / 499 k,
    // K = 100 605 k / 552 k / 450 k / 330 k, K = 200 230 k / 199 k / 149 k / 102 k).  16-bit counters: at least list / chunk_cap().
    int blocks_for(int nb, bool relaxed = false) const {
        int want = nblk;
        if (want <= 0) {
            const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
            const int64_t fill = (((int64_t)4 * cus) / std::max(nb * passes(), 1)) & ~(int64_t)7;
            want = (int)std::max<int64_t>(8, std::min<int64_t>(cus, fill));
        }
        if (const int64_t capn = chunk_cap(relaxed)) {
            const int64_t need = ceil_div(std::max<int64_t>(list_edges(), 1), capn);
            if (need > want) want = (int)(nblk > 0 ? need : ceil_div(need, 8) * 8);
        }
        return want;
    }
    int nblk_launch = 0;  // blocks per batch of the launch in flight (count -> reduce)
    int sym_launch = 0;   // k_reduce mode of the launch in flight (0 full edge list, 1 half list, 2 half list with self loops)
    int partial_blocks(int nb) const { return lds_path() ? blocks_for(nb) : 1; }
    size_t partial_words() const {  // largest nb * blocks_for(nb) * part_words over the launches this plan can issue
        size_t m = 0;
        for (int nb = 1; nb <= nbatch; ++nb) m = std::max(m, (size_t)nb * partial_blocks(nb));
        return m * part_words();
    }
    int resolve_tuning();
    int ensure_workspace(bool need_perms);
    int count_batches(int nb, int buf);  // slab[buf] -> partial for nb batches
    int reduce_batches(int nb, int64_t perm_batch0, int64_t perm_begin, int64_t perm_end, uint32_t* perms_out_dev);
};

int sqgr_nhood::resolve_tuning() {
    if (g && !wide()) SQGR_TRY(g->ensure_half());  // the counter mode (cm()) and with it every size below depend on the graph's symmetry
    if (wide()) {  // K*K*16 device-scope counters per batch: at most ~1 GiB of them in flight
        B = 16;
        const int64_t cap = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)K2 * 16 * 4));
        if (nbatch <= 0 || nbatch > cap) nbatch = (int)std::min<int64_t>(cap, 64);
    }
    if (B == 32 && (size_t)K2 * 32 * 4 > LDS_BUDGET) B = 16;
    if (B != 16 && B != 32) B = 16;
    const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    (void)cus;
    if (nbatch <= 0) {
        // 160 batches (2560 permutations) per launch group, fewer when two slab buffers of that size would not fit a
        // quarter of the free HBM (n up to 2^27 spots).  Measured at 1e6 spots x 30 clusters (tools/tune_sweep.sh, late round 2):
        // 64 batches 970 k permutations/s, 128: 1016 k, 160: 1020 k, 256: 1014 k, 320: 996 k — longer launches shorten the
        // ramp and the tail of the count kernel (0.466 -> 0.443 ms per 1000 permutations), beyond 256 the shuffle slows down.
        size_t free_b = 0, total_b = 0;
        nbatch = 160;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && n > 0) {
            const int64_t fit = (int64_t)(free_b / 4) / (2 * n * B);
            nbatch = (int)std::max<int64_t>(1, std::min<int64_t>(160, fit));
        }
    }
    return SQGR_OK;
}

int sqgr_nhood::ensure_workspace(bool need_perms) {
    SQGR_TRY(resolve_tuning());
    const size_t hw = (size_t)hist_words();
    SQGR_TRY(keys.ensure(2 * keys_stride()));
    SQGR_TRY(slab.ensure(2 * slab_stride()));
    for (int i = 0; i < 2; ++i) {
        if (!ev_shuffled[i]) SQGR_HIP(hipEventCreateWithFlags(&ev_shuffled[i], hipEventDisableTiming));
        if (!ev_counted[i]) SQGR_HIP(hipEventCreateWithFlags(&ev_counted[i], hipEventDisableTiming));
    }
    (void)hw;
    SQGR_TRY(partial.ensure(partial_words()));
    SQGR_TRY(acc_sum.ensure((size_t)nbatch * acc_words()));
    SQGR_TRY(acc_sq.ensure((size_t)nbatch * acc_words()));
    SQGR_TRY(shift.ensure((size_t)K2));
    SQGR_TRY(fin.ensure((size_t)2 * K2));
    (void)need_perms;
    return SQGR_OK;
}

int sqgr_nhood::count_batches(int nb, int buf) {
    const uint8_t* slab_p = slab.p + (size_t)buf * slab_stride();
    const int64_t nnz = g->nnz;
    hipStream_t st = ctx->stream;
    const int hw = hist_words();
    const int nblk = blocks_for(nb, columns_valid);  // shadows the tuning member on purpose: everything below is per launch
    nblk_launch = lds_path() ? nblk : 1;             // (the workspace is sized for partial_blocks(nb) >= nblk)
    sym_launch = 0;
    const int mode = cm();            // 0: 32-bit counters; 1 | 2: 16-bit counters (K*K | unordered pairs), reduced by k_reduce16
    const int ncell = cells();
    if (nnz == 0) {  // no edges: every count is zero
        SQGR_HIP(hipMemsetAsync(partial.p, 0, (size_t)nb * nblk_launch * part_words() * 4, st));
        return SQGR_OK;
    }
    if (wide()) {
        LaunchTimer t(ctx, "nhood_count_wide16");
        SQGR_HIP(hipMemsetAsync(partial.p, 0, (size_t)nb * hw * 4, st));
        k_count_wide16<<<dim3((unsigned)ceil_div(nnz, 256), nb), 256, 0, st>>>(nnz, g->erow.p, g->indices.p,
                                                                              reinterpret_cast<const uint16_t*>(slab_p), n, K, partial.p);
    } else if (lds_path() && B == 16 && be() < 16) {
        // 51 <= K <= 202: passes of be permutations over the (half) edge list, the passes of a chunk side by side on one XCD
        SQGR_TRY(g->ensure_half());
        const bool half = g->sym_state == 1;
        const uint32_t m = (uint32_t)(half ? g->n_half + g->n_self : nnz);
        const uint32_t self_begin = (uint32_t)(half ? g->n_half : nnz);
        const bool self = half && g->n_self > 0;
        sym_launch = self ? 2 : 0;
        const int e = be();
        // order of the list inside a wavefront's group of entries (sqgr_graph::pass_list): R = 1 — a gather instruction covers
        // consecutive edges — unless SQGR_COUNT_PASS_R says 2 or 4 (experiments; 4 = the list as it is)
        static const int order_r = [] { const char* v = getenv("SQGR_COUNT_PASS_R"); const int r = v ? atoi(v) : 1; return (r == 2 || r == 4) ? r : 1; }();
        // the list: 4 bytes per entry when the graph admits it (sqgr_graph::packed_list; SQGR_COUNT_PASS_PACK=0: never), else 8
        static const bool want_pack = [] { const char* v = getenv("SQGR_COUNT_PASS_PACK"); return !(v && atoi(v) == 0); }();
        const int2* list = nullptr;
        const uint32_t *plist = nullptr, *pbase = nullptr;
        // (measured at 1e6 spots, 2560 permutations per launch: 4 | 2 | 1 permutations per pass 2.33 -> 2.16, 7.46 -> 6.80, 7.88 -> 7.12 ms
        //  — and 8 per pass 2.07 -> 2.14 ms: two lanes per edge unpack every entry twice; that width keeps the 8-byte list)
        if (want_pack && e <= 4) SQGR_TRY(g->packed_list(64, order_r, &plist, &pbase));
        const bool pack = plist != nullptr;
        if (pack) list = reinterpret_cast<const int2*>(plist);
        else SQGR_TRY(g->pass_list(e >= 8 ? 32 : 64, order_r, e, &list));
        const uint32_t step = (uint32_t)(COUNT_THREADS * 4 / (e >= 8 ? 2 : 1));             // edges per iteration of a block
        const uint32_t epc = (uint32_t)(ceil_div(ceil_div((int64_t)m, nblk), step) * step);  // whole iterations per chunk
        const int sp = split();
        const size_t lds = mode ? (size_t)ncell * e * 2 : (size_t)((K + sp - 1) / sp) * K * e * 4;
        if (sp > 1 && half) sym_launch |= 4;  // h + h^T in k_reduce
        // (k_sum_chunks adds a batch's chunk partials in 32 bits: a pair's total, in doubled units with self loops, must fit — ADVICE r5)
        SQGR_REQUIRE(sp == 1 || (uint64_t)m * (self ? 2u : 1u) < ((uint64_t)1 << 32), "edge list too long for the 32-bit chunk sums of the split-row path (%u entries)", m);
        const int addt = (half && sp == 1) ? 1 : 0;
        const dim3 grid(nblk * (16 / e) * sp, nb);
        SQGR_REQUIRE(!mode || chunk_cap(columns_valid) == 0 || (int64_t)epc * (self ? 2 : 1) <= 65535, "internal: %u edges per block overflow a 16-bit counter", epc);
        SQGR_REQUIRE(mode != 2 || half, "internal: unordered-pair counters on a full edge list");
        LaunchTimer t(ctx, mode ? (half ? "nhood_count_pass16_half" : "nhood_count_pass16") : (half ? "nhood_count_pass_half" : "nhood_count_pass"));
#define SQGR_PASS_K(LPE, NS, SELF, SPLIT, PACK, CM)                                                                                  \
    do {                                                                                                                             \
        SQGR_TRY(allow_lds((k_count_pass<LPE, NS, SELF, SPLIT, PACK, CM>), lds));                                                    \
        k_count_pass<LPE, NS, SELF, SPLIT, PACK, CM><<<grid, COUNT_THREADS, lds, st>>>(m, list, pbase, slab_p, n, K, epc, self_begin, addt, nblk, \
                                                                                       (uint32_t)order_r, partial.p, ncell);        \
    } while (0)
#define SQGR_PASS_C(LPE, NS, SPLIT, CM)                                                    \
    do {                                                                                   \
        if (self) {                                                                        \
            if (pack) SQGR_PASS_K(LPE, NS, true, SPLIT, true, CM); else SQGR_PASS_K(LPE, NS, true, SPLIT, false, CM);   \
        } else {                                                                           \
            if (pack) SQGR_PASS_K(LPE, NS, false, SPLIT, true, CM); else SQGR_PASS_K(LPE, NS, false, SPLIT, false, CM); \
        }                                                                                  \
    } while (0)
#define SQGR_PASS(LPE, NS)                                  \
    do {                                                    \
        if (mode == 2) SQGR_PASS_C(LPE, NS, 1, 2);          \
        else if (mode == 1) SQGR_PASS_C(LPE, NS, 1, 1);     \
        else SQGR_PASS_C(LPE, NS, 1, 0);                    \
    } while (0)
        switch (e) {
            case 8: SQGR_PASS(2, 4); break;
            case 4: SQGR_PASS(1, 4); break;
            case 2: SQGR_PASS(1, 2); break;
            default:
                if (sp == 1) SQGR_PASS_C(1, 1, 1, 0); else SQGR_PASS_C(1, 1, 2, 0);
                break;
        }
#undef SQGR_PASS
#undef SQGR_PASS_C
#undef SQGR_PASS_K
    } else if (B == 32 || be() == 16) {
        // LDS-histogram kernels: on a structurally symmetric graph they walk the half list (see sqgr_graph::ensure_half)
        SQGR_TRY(g->ensure_half());
        const bool half = g->sym_state == 1;
        // a directed graph (K <= 50, 16 permutations per pass): its mutual pairs once + its edges without a mirror (ensure_split)
        bool split_list = false;
        if (!half && B == 16 && !mode && nblk >= 2) {
            SQGR_TRY(g->ensure_split());
            split_list = g->split_state == 1;
        }
        const int2* list = half ? g->half.p : (split_list ? g->split.p : g->coo.p);
        const uint32_t m = (uint32_t)(half ? g->n_half + g->n_self : (split_list ? g->n_mutual + g->n_oneway : nnz));
        const uint32_t self_begin = (uint32_t)(half ? g->n_half : (split_list ? g->n_mutual : nnz));
        const bool self = half && g->n_self > 0;
        sym_launch = self ? 2 : 0;  // the blocks' partials already hold h + h^T; half lists with self loops are in doubled units
        uint32_t epb = (uint32_t)(ceil_div(ceil_div((int64_t)m, nblk), 1024) * 1024);  // whole iterations of a block
        int addt = half ? 1 : 0;
        if (split_list) {  // the chunks are shared out in proportion to the two parts' lengths
            const int64_t M = g->n_mutual, O = g->n_oneway;
            const int nblk1 = (int)std::min<int64_t>(nblk - 1, std::max<int64_t>(1, (M * nblk + (M + O) / 2) / std::max<int64_t>(M + O, 1)));
            epb = (uint32_t)(ceil_div(std::max(ceil_div(M, nblk1), ceil_div(std::max<int64_t>(O, 1), nblk - nblk1)), 1024) * 1024);
            addt = 2 | (nblk1 << 2);
        }
        const size_t lds = mode ? (size_t)ncell * 32 : (size_t)hw * 4;
        const dim3 grid(nblk, nb);
        if (mode) {  // 51 <= K <= 100 (71 on directed graphs): 16-bit counters, all 16 permutations in one pass
            SQGR_REQUIRE(chunk_cap(columns_valid) == 0 || (int64_t)epb * (self ? 2 : 1) <= 65535, "internal: %u edges per block overflow a 16-bit counter", epb);
            SQGR_REQUIRE(mode != 2 || half, "internal: unordered-pair counters on a full edge list");
            const int pw16 = ncell * 8;  // 32-bit words of a block's partial = its LDS image
            LaunchTimer t(ctx, half ? "nhood_count_c16_half" : "nhood_count_c16");
#define SQGR_COUNT16(MW, SELF, CM)                                                                                             \
    do {                                                                                                                       \
        SQGR_TRY(allow_lds((k_count<16, MW, SELF, true, 0, CM>), lds));                                                        \
        k_count<16, MW, SELF, true, 0, CM><<<grid, COUNT_THREADS, lds, st>>>(m, list, slab_p, n, K, pw16, epb, self_begin, 0, partial.p); \
    } while (0)
#define SQGR_COUNT16_S(MW, CM) \
    do { if (self) SQGR_COUNT16(MW, true, CM); else SQGR_COUNT16(MW, false, CM); } while (0)
            if (lds * 2 <= LDS_BUDGET) {
                if (mode == 2) SQGR_COUNT16_S(8, 2); else SQGR_COUNT16_S(8, 1);
            } else {
                if (mode == 2) SQGR_COUNT16_S(4, 2); else SQGR_COUNT16_S(4, 1);
            }
#undef SQGR_COUNT16_S
#undef SQGR_COUNT16
            SQGR_HIP(hipGetLastError());
            return SQGR_OK;
        }
#define SQGR_COUNT(BB, MW, SELF) \
    k_count<BB, MW, SELF><<<grid, COUNT_THREADS, lds, st>>>(m, list, slab_p, n, K, hw, epb, self_begin, addt, partial.p)
#define SQGR_COUNT_D(BB, MW, SELF) \
    k_count<BB, MW, SELF, true><<<grid, COUNT_THREADS, lds, st>>>(m, list, slab_p, n, K, hw, epb, self_begin, addt, partial.p)
        static const bool dot2 = [] { const char* e = getenv("SQGR_COUNT_DOT2"); return !(e && atoi(e) == 0); }();
        if (B == 32) {
            LaunchTimer t(ctx, half ? "nhood_count_b32_half" : "nhood_count_b32");
            if (self) {
                SQGR_TRY(allow_lds(k_count<32, 4, true>, lds));
                SQGR_COUNT(32, 4, true);
            } else {
                SQGR_TRY(allow_lds(k_count<32, 4, false>, lds));
                SQGR_COUNT(32, 4, false);
            }
        } else {
            LaunchTimer t(ctx, half ? "nhood_count_b16_half" : (split_list ? "nhood_count_b16_split" : "nhood_count_b16"));
            static const int dbg = [] { const char* e = getenv("SQGR_COUNT_DEBUG"); return e ? atoi(e) : 0; }();
            if (dot2 && dbg && lds * 2 <= LDS_BUDGET && !self) {
#define SQGR_COUNT_DBG(D) \
    case D: k_count<16, 8, false, true, D><<<grid, COUNT_THREADS, lds, st>>>(m, list, slab_p, n, K, hw, epb, self_begin, addt, partial.p); break
                switch (dbg) {
                    SQGR_COUNT_DBG(1); SQGR_COUNT_DBG(2); SQGR_COUNT_DBG(3); SQGR_COUNT_DBG(4); SQGR_COUNT_DBG(5); SQGR_COUNT_DBG(6);
                    default: k_count<16, 8, false, true, 7><<<grid, COUNT_THREADS, lds, st>>>(m, list, slab_p, n, K, hw, epb, self_begin, addt, partial.p);
                }
#undef SQGR_COUNT_DBG
            } else if (dot2) {
                if (lds * 2 <= LDS_BUDGET) {
                    if (self) SQGR_COUNT_D(16, 8, true); else SQGR_COUNT_D(16, 8, false);
                } else if (self) {
                    SQGR_TRY(allow_lds(k_count<16, 4, true, true>, lds));
                    SQGR_COUNT_D(16, 4, true);
                } else {
                    SQGR_TRY(allow_lds(k_count<16, 4, false, true>, lds));
                    SQGR_COUNT_D(16, 4, false);
                }
            } else if (lds * 2 <= LDS_BUDGET) {
                if (self) SQGR_COUNT(16, 8, true); else SQGR_COUNT(16, 8, false);
            } else if (self) {
                SQGR_TRY(allow_lds(k_count<16, 4, true>, lds));
                SQGR_COUNT(16, 4, true);
            } else {
                SQGR_TRY(allow_lds(k_count<16, 4, false>, lds));
                SQGR_COUNT(16, 4, false);
            }
        }
#undef SQGR_COUNT
#undef SQGR_COUNT_D
    } else {  // K*K counters do not fit LDS: device-scope atomics into ONE histogram per batch
        const int gblk = std::max(nblk, 64);
        const int64_t epb = ceil_div(nnz, gblk);
        LaunchTimer t(ctx, "nhood_count_global");
        SQGR_HIP(hipMemsetAsync(partial.p, 0, (size_t)nb * hw * 4, st));
        k_count_global<<<dim3(gblk, nb), COUNT_THREADS, 0, st>>>(nnz, g->erow.p, g->indices.p, slab_p, n, K, epb, partial.p);
    }
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

int sqgr_nhood::reduce_batches(int nb, int64_t perm_batch0, int64_t perm_begin, int64_t perm_end, uint32_t* perms_out_dev) {
    const int hw = hist_words();
    LaunchTimer t(ctx, "nhood_reduce");
    if (const int mode = cm()) {  // 16-bit partials: [planes][cell][pw] -> one accumulator slot per ordered pair and batch
        const int ncell = cells(), pw = partial_w();
        const dim3 grid((unsigned)ceil_div(ncell, 256), nb);
        const int dbl = (sym_launch & 2) ? 1 : 0;
#define SQGR_RED16(TRI, PW)                                                                                                                    \
    k_reduce16<TRI, PW><<<grid, 256, 0, ctx->stream>>>(partial.p, nblk_launch, ncell, K, dbl, shift.p, perm_batch0, perm_begin, perm_end, acc_sum.p, \
                                                       acc_sq.p, perms_out_dev)
#define SQGR_RED16_W(TRI)                                \
    switch (pw) {                                        \
        case 16: SQGR_RED16(TRI, 16); break;             \
        case 8: SQGR_RED16(TRI, 8); break;               \
        case 4: SQGR_RED16(TRI, 4); break;               \
        default: SQGR_RED16(TRI, 2); break;              \
    }
        if (mode == 2) { SQGR_RED16_W(true) } else { SQGR_RED16_W(false) }
#undef SQGR_RED16_W
#undef SQGR_RED16
        SQGR_HIP(hipGetLastError());
        return SQGR_OK;
    }
    int nsum = nblk_launch;
    if ((sym_launch & 4) && nblk_launch > 1) {  // the transposed reads are scattered: do them once per slot, not once per chunk
        k_sum_chunks<<<dim3((unsigned)ceil_div(hw, 256), nb), 256, 0, ctx->stream>>>(partial.p, nblk_launch, hw);
        nsum = 1;
    }
    k_reduce<<<dim3((unsigned)ceil_div(hw, 64), nb), 256, 0, ctx->stream>>>(partial.p, nblk_launch, nsum, hw, B, partial_w(), K, sym_launch, shift.p,
                                                                            perm_batch0, perm_begin, perm_end, acc_sum.p,
                                                                            acc_sq.p, perms_out_dev);
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

static int check_labels(const int32_t* labels, int64_t n, int K, bool allow_negative) {
    for (int64_t i = 0; i < n; ++i) {
        if (labels[i] >= K || (labels[i] < 0 && !allow_negative)) {
            set_error("labels[%lld]=%d outside [0,%d)", (long long)i, labels[i], K);
            return SQGR_ERR_INVALID;
        }
    }
    return SQGR_OK;
}

static int edge_pairs(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, bool weighted, bool allow_negative,
                      unsigned long long* out_u64, double* out_f64) {
    SQGR_REQUIRE(ctx && g && labels, "ctx/graph/labels is NULL");
    SQGR_REQUIRE(g->ctx == ctx, "graph belongs to a different context");
    SQGR_REQUIRE(K >= 1, "K=%d", K);
    SQGR_TRY(check_labels(labels, g->n, K, allow_negative));
    SQGR_HIP(hipSetDevice(ctx->device));
    DevBuf<int32_t> dl;
    DevBuf<unsigned long long> du;
    DevBuf<double> dd;
    const size_t K2 = (size_t)K * K;
    SQGR_TRY(dl.alloc((size_t)g->n));
    SQGR_HIP(hipMemcpyAsync(dl.p, labels, (size_t)g->n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (weighted) {
        SQGR_REQUIRE(g->has_data || g->nnz == 0, "graph was uploaded without edge data; weights unavailable");
        SQGR_TRY(dd.alloc(K2));
        SQGR_HIP(hipMemsetAsync(dd.p, 0, K2 * 8, ctx->stream));
    } else {
        SQGR_TRY(du.alloc(K2));
        SQGR_HIP(hipMemsetAsync(du.p, 0, K2 * 8, ctx->stream));
    }
    if (g->nnz > 0) {
        LaunchTimer t(ctx, "nhood_edge_pairs");
        unsigned grid = (unsigned)ceil_div(g->nnz, 256);
        if (weighted) {
            const size_t cell_bytes = K2 * 8;
            const int waves = cell_bytes * 4 <= 64 * 1024 ? 4 : (cell_bytes <= LDS_BUDGET ? 1 : 0);
            if (waves == 0) {  // K > 143: the accumulator of one wave does not fit LDS
                k_edge_pairs<true><<<grid, 256, 0, ctx->stream>>>(g->nnz, g->erow.p, g->indices.p, g->data.p, dl.p, K, nullptr, dd.p);
            } else {
                const int64_t epw = std::max<int64_t>(64, ceil_div(ceil_div(g->nnz, 4096), 64) * 64);
                const int64_t nwaves = ceil_div(ceil_div(g->nnz, epw), waves) * waves;
                DevBuf<double> parts;
                SQGR_TRY(parts.alloc((size_t)nwaves * K2));
                if (waves == 4) {
                    k_edge_weight_partials<4><<<(unsigned)(nwaves / 4), 256, cell_bytes * 4, ctx->stream>>>(g->nnz, g->erow.p, g->indices.p, g->data.p,
                                                                                                           dl.p, K, epw, parts.p);
                } else {
                    SQGR_TRY(allow_lds(k_edge_weight_partials<1>, cell_bytes));
                    k_edge_weight_partials<1><<<(unsigned)nwaves, 64, cell_bytes, ctx->stream>>>(g->nnz, g->erow.p, g->indices.p, g->data.p, dl.p, K,
                                                                                                epw, parts.p);
                }
                k_sum_partials<<<(unsigned)ceil_div((int64_t)K2, 64), 64, 0, ctx->stream>>>(parts.p, nwaves, (int)K2, dd.p);
                SQGR_HIP(hipGetLastError());
                SQGR_HIP(hipStreamSynchronize(ctx->stream));  // `parts` is released on return
            }
        }
        else
            k_edge_pairs<false><<<grid, 256, 0, ctx->stream>>>(g->nnz, g->erow.p, g->indices.p, nullptr, dl.p, K, du.p, nullptr);
        SQGR_HIP(hipGetLastError());
    }
    if (weighted)
        SQGR_HIP(hipMemcpyAsync(out_f64, dd.p, K2 * 8, hipMemcpyDeviceToHost, ctx->stream));
    else
        SQGR_HIP(hipMemcpyAsync(out_u64, du.p, K2 * 8, hipMemcpyDeviceToHost, ctx->stream));
    SQGR_HIP(hipStreamSynchronize(ctx->stream));
    return SQGR_OK;
}

extern "C" {

int sqgr_nhood_counts(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, uint32_t* out_counts) {
    SQGR_REQUIRE(out_counts, "out_counts is NULL");
    SQGR_REQUIRE(K >= 2, "Expected at least `2` clusters, found `%d`.", K);  // gr/_nhood.py:107-108
    std::vector<unsigned long long> tmp((size_t)K * K);
    SQGR_TRY(edge_pairs(ctx, g, labels, K, false, false, tmp.data(), nullptr));
    for (size_t i = 0; i < tmp.size(); ++i) out_counts[i] = (uint32_t)tmp[i];  // uint32 like the reference (wraps)
    return SQGR_OK;
}

int sqgr_interaction_matrix(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, int32_t weights,
                            double* out) {
    SQGR_REQUIRE(out, "out is NULL");
    if (weights) return edge_pairs(ctx, g, labels, K, true, true, nullptr, out);
    std::vector<unsigned long long> tmp((size_t)K * K);
    SQGR_TRY(edge_pairs(ctx, g, labels, K, false, true, tmp.data(), nullptr));
    for (size_t i = 0; i < tmp.size(); ++i) out[i] = (double)tmp[i];
    return SQGR_OK;
}

// Builds the label tables of a plan.  `g` may be NULL (label shuffling only: the ligrec path), then `n` is the number of
// labelled items.
static int nhood_build(sqgr_ctx* ctx, const sqgr_graph* g, int64_t n, const int32_t* labels, int32_t K, const int32_t* lib_ids,
                       int32_t n_libs, sqgr_nhood** out_plan) {
    SQGR_REQUIRE(ctx && out_plan, "ctx/out_plan is NULL");
    *out_plan = nullptr;
    SQGR_REQUIRE(!g || g->ctx == ctx, "graph belongs to a different context");
    SQGR_REQUIRE(K >= 2, "Expected at least `2` clusters, found `%d`.", K);
    // with a graph: K*K*16 device-scope counters per batch bound the cluster count; the label generators alone (ligrec's
    // shuffler: no graph) address 16-bit labels
    if (K > (g ? 4096 : 65535)) {  // (4096 clusters: 1 GiB of counters per batch of 16 permutations, 2 x 2 GiB of accumulator slots)
        set_error(g ? "K=%d > 4096 clusters is not supported by the batched permutation kernels"
                    : "K=%d > 65535 clusters: the label generators write 16-bit labels", K);
        return SQGR_ERR_UNSUPPORTED;
    }
    if (n > (int64_t)1 << 27 || (g && g->nnz > (int64_t)0xFFF00000u)) {
        set_error("graph too large for 32-bit slab offsets (n=%lld, nnz=%lld)", (long long)n, (long long)(g ? g->nnz : 0));
        return SQGR_ERR_UNSUPPORTED;
    }
    if (labels) SQGR_TRY(check_labels(labels, n, K, false));
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_nhood* p = new sqgr_nhood();
    p->ctx = ctx;
    p->g = g;
    p->n = n;
    p->K = K;
    p->K2 = K * K;
    p->dom0.dom = make_domain((uint32_t)n);
    p->dom0.aoff = 0;
    int rc = SQGR_OK;
    do {
        const bool libs_on = lib_ids && n_libs >= 1;
        p->has_libs = libs_on;
        p->n_libs = libs_on ? n_libs : 1;
        p->kpad = K + 1;  // K boundaries + UINT_MAX sentinel
        // label histogram per library -> boundaries cum[l][k] = #{members of l with label < k}, padded with UINT_MAX
        std::vector<int64_t> cnt((size_t)p->n_libs, 0);
        std::vector<uint32_t> hist((size_t)p->n_libs * K, 0);
        std::vector<int32_t> rank;
        if (libs_on) rank.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            int l = 0;
            if (libs_on) {
                l = lib_ids[i];
                if (l < 0 || l >= n_libs) {
                    set_error("lib_ids[%lld]=%d outside [0,%d)", (long long)i, l, n_libs);
                    rc = SQGR_ERR_INVALID;
                    break;
                }
                rank[i] = (int32_t)cnt[l];  // ranks in position order == np.where(libraries == c)[0]
            }
            cnt[l]++;
            if (labels) hist[(size_t)l * K + labels[i]]++;
        }
        if (rc != SQGR_OK) break;
        if (labels) {
            for (int k = 0; k < K; ++k) {
                int64_t tot = 0;
                for (int l = 0; l < p->n_libs; ++l) tot += hist[(size_t)l * K + k];
                p->max_label_count = (int)std::max<int64_t>(p->max_label_count, tot);
            }
        }
        std::vector<uint32_t> cum((size_t)p->n_libs * p->kpad, 0xFFFFFFFFu);
        std::vector<LibDom> doms((size_t)p->n_libs);
        size_t blk_total = 0;
        for (int l = 0; l < p->n_libs; ++l) {
            uint32_t run = 0;
            for (int k = 0; k < K; ++k) {
                cum[(size_t)l * p->kpad + k] = run;
                run += hist[(size_t)l * K + k];
            }
            doms[l].dom = make_domain((uint32_t)(cnt[l] > 0 ? cnt[l] : 1));
            doms[l].aoff = (uint32_t)blk_total;
            blk_total += doms[l].dom.A;
        }
        p->blk_bytes = (int)blk_total;
        // the boundary table rides in LDS next to the block table when both fit; the 16-bit generator (more than 256 labels)
        // otherwise reads the boundaries of its exact route from global memory (thousands of clusters: ligrec's shuffler)
        p->tab_lds = (size_t)p->n_libs * p->kpad * 4 + blk_total * 4 <= 150 * 1024;
        if (!p->tab_lds && (K <= 256 || blk_total * 4 > 150 * 1024)) {
            set_error("library/label tables (%zu bytes) exceed the LDS budget of the shuffle kernel",
                      (size_t)p->n_libs * p->kpad * 4 + blk_total * 4);
            rc = SQGR_ERR_UNSUPPORTED;
            break;
        }
        // block table (appended to the boundary table): one word per high digit a describing the ranks [a*B, (a+1)*B) of
        // the label-sorted base — byte 0 the label of rank a*B, bits 16-31 the low digit at which the NEXT label starts
        // (0xFFFF: none).  "Events" inside a block are label starts and the library's end (ranks >= n_l read as label K,
        // the sentinel that sends the kernel to its exact route); a block with one event whose label is lab + 1 fits the
        // two-field form, a block with no event too, everything else is marked with lab0 = K.
        std::vector<uint32_t> blk(blk_total, 0);
        for (int l = 0; l < p->n_libs; ++l) {
            const uint32_t* c = &cum[(size_t)l * p->kpad];
            const uint64_t n_l = (uint64_t)cnt[l];
            const uint32_t SENT = (uint32_t)K | (0xFFFFu << 16);  // bits 0-15: label K (byte 0 = K for K <= 255; K = 256 reads 0 + bit 8)
            uint32_t lab = 0;
            for (uint32_t a = 0; a < doms[l].dom.A; ++a) {
                const uint64_t lo = (uint64_t)a * doms[l].dom.B, hi = lo + doms[l].dom.B;
                uint32_t word;
                if (lo >= n_l) {
                    word = SENT;
                } else {
                    while (lab + 1 < (uint32_t)K && c[lab + 1] <= lo) ++lab;  // label of rank lo: the largest k with c[k] <= lo
                    const uint64_t end = hi < n_l ? hi : n_l;
                    uint32_t events = 0, ev_label = 0;
                    uint64_t ev_pos = 0;
                    for (uint32_t k2 = lab + 1; k2 < (uint32_t)K && c[k2] < end; ++k2) {  // label changes inside (lo, end)
                        const uint64_t nxt = k2 + 1 < (uint32_t)K ? c[k2 + 1] : n_l;
                        if (c[k2] >= nxt) continue;  // empty category: no rank carries it
                        if (events == 0) { ev_label = k2; ev_pos = c[k2] - lo; }
                        ++events;
                    }
                    if (hi > n_l) {  // the library ends inside this block: ranks >= n_l read as the sentinel label K
                        if (events == 0) { ev_label = (uint32_t)K; ev_pos = n_l - lo; }
                        ++events;
                    }
                    if (events == 0)
                        word = lab | (0xFFFFu << 16);
                    else if (events == 1 && ev_label == lab + 1 && ev_pos > 0)
                        word = lab | ((uint32_t)ev_pos << 16);
                    else  // the exact route ranks x against the boundaries: from this block's first label on (byte 1: free in the
                        word = SENT | (K <= 255 ? lab << 8 : 0u);  // two-field form, which reads byte 0 and the upper half)
                }
                blk[doms[l].aoff + a] = word;
            }
        }
        const size_t cum_words = cum.size();
        cum.insert(cum.end(), blk.begin(), blk.end());
        (void)cum_words;
        if (!libs_on) p->dom0 = doms[0];
        {   // numpy-compatible mode: labels in library-grouped position order + library offsets
            std::vector<uint32_t> off((size_t)p->n_libs + 1, 0);
            for (int l = 0; l < p->n_libs; ++l) off[l + 1] = off[l] + (uint32_t)cnt[l];
            std::vector<uint8_t> base((size_t)n, 0);
            if (labels)
                for (int64_t i = 0; i < n; ++i) base[libs_on ? off[lib_ids[i]] + (uint32_t)rank[i] : (uint32_t)i] = (uint8_t)labels[i];
            p->has_labels = labels != nullptr;
            if (K > 256 && labels) {  // 16-bit copy of the same vector
                std::vector<uint16_t> b16((size_t)n, 0);
                for (int64_t i = 0; i < n; ++i) b16[libs_on ? off[lib_ids[i]] + (uint32_t)rank[i] : (uint32_t)i] = (uint16_t)labels[i];
                if ((rc = p->base16.alloc((size_t)n)) != SQGR_OK) break;
                if (hipMemcpy(p->base16.p, b16.data(), (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess) { rc = SQGR_ERR_HIP; break; }
            }
            if ((rc = p->base_pos.alloc((size_t)n)) != SQGR_OK) break;
            if ((rc = p->lib_off.alloc(off.size())) != SQGR_OK) break;
            hipError_t e2 = hipMemcpy(p->base_pos.p, base.data(), (size_t)n, hipMemcpyHostToDevice);
            if (e2 == hipSuccess) e2 = hipMemcpy(p->lib_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice);
            if (e2 != hipSuccess) {
                set_error("label upload failed: %s", hipGetErrorString(e2));
                rc = SQGR_ERR_HIP;
                break;
            }
        }
        if ((rc = p->cum.alloc(cum.size())) != SQGR_OK) break;
        hipError_t e = hipMemcpy(p->cum.p, cum.data(), cum.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && libs_on) {
            if ((rc = p->lib_of.alloc((size_t)n)) != SQGR_OK) break;
            if ((rc = p->rank_of.alloc((size_t)n)) != SQGR_OK) break;
            if ((rc = p->libs.alloc((size_t)n_libs)) != SQGR_OK) break;
            e = hipMemcpy(p->lib_of.p, lib_ids, (size_t)n * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->rank_of.p, rank.data(), (size_t)n * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->libs.p, doms.data(), (size_t)n_libs * sizeof(LibDom), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            set_error("label upload failed: %s", hipGetErrorString(e));
            rc = SQGR_ERR_HIP;
        }
    } while (0);
    if (rc != SQGR_OK) {
        delete p;
        return rc;
    }
    *out_plan = p;
    return SQGR_OK;
}

int sqgr_nhood_create(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, const int32_t* lib_ids,
                      int32_t n_libs, sqgr_nhood** out_plan) {
    SQGR_REQUIRE(ctx && g && out_plan, "ctx/graph/out_plan is NULL");
    return nhood_build(ctx, g, g->n, labels, K, lib_ids, n_libs, out_plan);
}

int sqgr_nhood_destroy(sqgr_nhood* plan) {
    if (!plan) return SQGR_OK;
    (void)hipSetDevice(plan->ctx->device);
    delete plan;
    return SQGR_OK;
}

int sqgr_nhood_tune(sqgr_nhood* plan, int32_t perms_per_pass, int32_t blocks_per_batch, int32_t batches_per_launch) {
    SQGR_REQUIRE(plan, "plan is NULL");
    SQGR_REQUIRE(perms_per_pass == 0 || perms_per_pass == 16 || perms_per_pass == 32 || perms_per_pass == 8 || perms_per_pass == 4 ||
                     perms_per_pass == 2 || perms_per_pass == 1,
                 "perms_per_pass must be 0, 32, 16 (slab width) or 8, 4, 2, 1 (cap of the LDS pass width of a 16-wide slab)");
    SQGR_REQUIRE(blocks_per_batch >= 0 && blocks_per_batch <= 65535 && batches_per_launch >= 0 && batches_per_launch <= 1024,
                 "tuning value out of range");
    plan->B = perms_per_pass == 32 ? 32 : 16;
    plan->pass_cap = (perms_per_pass > 0 && perms_per_pass < 16) ? perms_per_pass : 16;
    plan->nblk = blocks_per_batch;
    plan->nbatch = batches_per_launch;  // 0: automatic (resolve_tuning)
    // force re-allocation with the new geometry
    plan->keys.release(); plan->slab.release(); plan->partial.release(); plan->acc_sum.release(); plan->acc_sq.release();
    return SQGR_OK;
}

static int launch_shuffle_raw(sqgr_nhood* p, int B, int nb, const uint32_t* keys, uint8_t* slab, hipStream_t st, int pw = 16) {
    unsigned gx = (unsigned)ceil_div(p->n, 256);
    // ~96 blocks per CU over all batches of the launch, each walking several spots (grid stride): amortises the LDS table
    // set-up and the key loads — measured 7 % faster on MI355X than one block per 256 spots (tools/shuf_sweep.sh)
    int per_cu = 96;
    const char* env_blocks = getenv("SQGR_SHUFFLE_BLOCKS_PER_CU");
    if (env_blocks && atoi(env_blocks) > 0) per_cu = atoi(env_blocks);
    gx = std::min<unsigned>(gx, (unsigned)(per_cu * std::max(p->ctx->cu_count, 1)) / (unsigned)std::max(nb, 1) + 1);
    LaunchTimer t(p->ctx, p->wide() ? "nhood_shuffle16" : "nhood_shuffle", st);
    const size_t lds = (p->tab_lds ? (size_t)p->n_libs * p->kpad * 4 : 0) + (size_t)p->blk_bytes * 4;
    if (p->wide()) {
        if (B != 16) {
            set_error("more than 256 clusters: 16 permutations per pass only");
            return SQGR_ERR_UNSUPPORTED;
        }
        uint16_t* slab16 = reinterpret_cast<uint16_t*>(slab);
        if (p->has_libs) {
            SQGR_TRY(allow_lds(k_shuffle16<true>, lds));
            k_shuffle16<true><<<dim3(gx, nb), 256, lds, st>>>(p->n, p->cum.p, p->kpad, p->blk_bytes, p->K, keys, p->dom0, p->n_libs, p->lib_of.p,
                                                              p->rank_of.p, p->libs.p, slab16, p->tab_lds ? 1 : 0);
        } else {
            SQGR_TRY(allow_lds(k_shuffle16<false>, lds));
            k_shuffle16<false><<<dim3(gx, nb), 256, lds, st>>>(p->n, p->cum.p, p->kpad, p->blk_bytes, p->K, keys, p->dom0, p->n_libs, p->lib_of.p,
                                                               p->rank_of.p, p->libs.p, slab16, p->tab_lds ? 1 : 0);
        }
        SQGR_HIP(hipGetLastError());
        return SQGR_OK;
    }
    if (p->independent) {  // every permutation its own 8-round bijection (bench.py's `nhood_independent_bijections`; keys: k_keygen_indep)
        if (B != 16 || p->has_libs) {
            set_error("SQGR_SHUFFLE_INDEPENDENT: 16 permutations per row and no libraries only");
            return SQGR_ERR_UNSUPPORTED;
        }
        if (p->K <= 126) k_shuffle_indep<true><<<dim3(gx, nb), 256, lds, st>>>(p->n, p->cum.p, p->kpad, p->blk_bytes, p->K, keys, p->dom0, slab, pw);
        else k_shuffle_indep<false><<<dim3(gx, nb), 256, lds, st>>>(p->n, p->cum.p, p->kpad, p->blk_bytes, p->K, keys, p->dom0, slab, pw);
        SQGR_HIP(hipGetLastError());
        return SQGR_OK;
    }
    const unsigned gy = (unsigned)(B == 32 ? nb : (nb + 1) / 2);  // 16-permutation rows are shuffled in pairs
#define SQGR_SHUFFLE(BB, LIBS, SK)                                                                                               \
    k_shuffle<BB, LIBS, SK><<<dim3(gx, gy), 256, lds, st>>>(p->n, p->cum.p, p->kpad, p->blk_bytes, p->K, keys, p->dom0, p->n_libs, nb, \
                                                            p->lib_of.p, p->rank_of.p, p->libs.p, slab, pw, p->spot_of.p)
#define SQGR_SHUFFLE_K(BB, LIBS) \
    if (p->K <= 126) SQGR_SHUFFLE(BB, LIBS, true); else SQGR_SHUFFLE(BB, LIBS, false)
    if (B == 32) {
        if (p->has_libs) { SQGR_SHUFFLE_K(32, true); } else { SQGR_SHUFFLE_K(32, false); }
    } else {
        if (p->has_libs) { SQGR_SHUFFLE_K(16, true); } else { SQGR_SHUFFLE_K(16, false); }
    }
#undef SQGR_SHUFFLE_K
#undef SQGR_SHUFFLE
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

static int launch_shuffle(sqgr_nhood* p, int nb, int buf, hipStream_t st) {
    return launch_shuffle_raw(p, p->B, nb, p->keys.p + (size_t)buf * p->keys_stride(), p->slab.p + (size_t)buf * p->slab_stride(), st, p->plane_w());
}

int sqgr_nhood_run(sqgr_nhood* plan, uint64_t seed, int64_t perm_begin, int64_t perm_end, const int64_t* shift,
                   int64_t* out_sum, uint64_t* out_sumsq, uint32_t* out_perms) {
    SQGR_REQUIRE(plan && out_sum && out_sumsq, "plan/out_sum/out_sumsq is NULL");
    SQGR_REQUIRE(perm_begin >= 0 && perm_end >= perm_begin, "bad permutation range [%lld,%lld)", (long long)perm_begin,
                 (long long)perm_end);
    sqgr_nhood* p = plan;
    sqgr_ctx* ctx = p->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int K2 = p->K2;
    const int64_t nperm = perm_end - perm_begin;
    // this rank's own part (everything in front of the collective); with a communicator its outcome is agreed on first, so
    // that a rank that fails here never leaves its peers waiting inside the all-reduce
    {   // SQGR_SHUFFLE_INDEPENDENT=1: the generator without its shared group bijection (read at every call; see k_shuffle_indep)
        const char* e = getenv("SQGR_SHUFFLE_INDEPENDENT");
        p->independent = e && atoi(e) == 1;
        SQGR_REQUIRE(!p->independent || (p->B == 16 && !p->has_libs && !p->wide()), "SQGR_SHUFFLE_INDEPENDENT needs 16-wide rows, K <= 256 and no libraries");
        SQGR_REQUIRE(!p->independent || !p->mapped(), "SQGR_SHUFFLE_INDEPENDENT: no spot map");
    }
    auto local = [&]() -> int {
    SQGR_TRY(p->ensure_workspace(out_perms != nullptr));
    const int B = p->B, hw = p->acc_words();
    const int64_t per_launch = (int64_t)p->nbatch * B;
    if (shift)
        SQGR_HIP(hipMemcpyAsync(p->shift.p, shift, (size_t)K2 * 8, hipMemcpyHostToDevice, st));
    else
        SQGR_HIP(hipMemsetAsync(p->shift.p, 0, (size_t)K2 * 8, st));
    SQGR_HIP(hipMemsetAsync(p->acc_sum.p, 0, (size_t)p->nbatch * hw * 8, st));
    SQGR_HIP(hipMemsetAsync(p->acc_sq.p, 0, (size_t)p->nbatch * hw * 8, st));
    if (out_perms && nperm > 0) SQGR_TRY(p->perms_dev.ensure((size_t)nperm * K2));
    // Two-stage pipeline on two streams: shuffles (VALU-bound) of launch group g+1 overlap the counting
    // (LDS-atomic/memory-bound) of group g; slab and key buffers ping-pong, ordered by events.
    // (measured on MI355X: both stages are VALU-issue bound, so overlapping them gains ~0-3 %; the default keeps one
    //  stream so per-kernel timings stay exclusive — set SQGR_NHOOD_STREAMS=2 to overlap)
    const char* env_streams = getenv("SQGR_NHOOD_STREAMS");
    hipStream_t sa = (env_streams && atoi(env_streams) == 2) ? ctx->stream2 : st;
    int64_t grp = 0;
    // the label generator works in groups of 16 permutations (sqgr_rng.h): rows start at multiples of 16 of the GLOBAL
    // permutation index, permutations in front of perm_begin are generated and masked out by k_reduce
    for (int64_t p0 = perm_begin - perm_begin % FEISTEL_GROUP; p0 < perm_end; p0 += per_launch, ++grp) {
        const int64_t todo = (perm_end - p0 < per_launch) ? perm_end - p0 : per_launch;
        const int nb = (int)ceil_div(todo, B);
        // one stream: the kernels of a group run in order, ONE slab serves every group (a launch group sized to the Infinity
        // Cache then never alternates between two of them); two streams: slab and keys ping-pong
        const int buf = (sa != st) ? (int)(grp & 1) : 0;
        if (sa != st && grp >= 2) SQGR_HIP(hipStreamWaitEvent(sa, p->ev_counted[buf], 0));  // slab[buf] has been consumed
        {
            LaunchTimer t(ctx, "nhood_keygen", sa);
            const int64_t nk = (int64_t)nb * p->n_libs * (B / FEISTEL_GROUP + B / 2);
            if (p->independent)
                k_keygen_indep<<<(unsigned)ceil_div((int64_t)nb * 8, 256), 256, 0, sa>>>(seed, p0, nb, p->keys.p + (size_t)buf * p->keys_stride());
            else
                k_keygen<<<(unsigned)ceil_div(nk, 256), 256, 0, sa>>>(seed, p0, nb, B, p->n_libs, p->keys.p + (size_t)buf * p->keys_stride());
            SQGR_HIP(hipGetLastError());
        }
        SQGR_TRY(launch_shuffle(p, nb, buf, sa));
        SQGR_HIP(hipEventRecord(p->ev_shuffled[buf], sa));
        SQGR_HIP(hipStreamWaitEvent(st, p->ev_shuffled[buf], 0));
        p->columns_valid = p->has_labels;  // every column of every row is a shuffle of the base labels (see chunk_cap)
        const int rc_count = p->count_batches(nb, buf);
        p->columns_valid = false;
        SQGR_TRY(rc_count);
        SQGR_TRY(p->reduce_batches(nb, p0, perm_begin, perm_end, out_perms ? p->perms_dev.p : nullptr));
        SQGR_HIP(hipEventRecord(p->ev_counted[buf], st));
    }
    {
        LaunchTimer t(ctx, "nhood_finalize");
        SQGR_HIP(hipMemsetAsync(p->fin.p, 0, (size_t)2 * K2 * 8, st));
        k_finalize<<<(unsigned)ceil_div(hw, 256), 256, 0, st>>>(p->acc_sum.p, p->acc_sq.p, p->nbatch, hw, p->acc_pw(), K2, p->fin.p,
                                                               reinterpret_cast<uint64_t*>(p->fin.p + K2));
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
    };
    // (The swap records of the bucketed replay — up to 36 GB at 1e6 spots x 8192 permutations per pass — stay with the plan's
    // workspace until the plan goes: releasing them after every run was tried for ADVICE r4 and made a persistent plan re-allocate
    // 35 GB per run, more than the parked-buffer pool keeps; the driver then took seconds for some of those hipMallocs
    // (bench.py's numpy leg: 61 k -> 3.9 k permutations/s in one lease).  The front ends create and close their plan per call, so
    // a call gives the memory back either way; a caller that keeps a plan decides with sqgr_nhood_destroy / sqgr_ctx_trim.)
    SQGR_TRY(comm_agree(p->comm, local(), st));
    // multi-GPU: the ranks ran disjoint permutation ranges; one RCCL all-reduce of the 2*K*K exact integer moments on the
    // device, then every rank copies out the global sums
    SQGR_TRY(comm_allreduce_i64_dev(p->comm, p->fin.p, (size_t)2 * K2, false, st));
    SQGR_HIP(hipMemcpyAsync(out_sum, p->fin.p, (size_t)K2 * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipMemcpyAsync(out_sumsq, p->fin.p + K2, (size_t)K2 * 8, hipMemcpyDeviceToHost, st));
    if (out_perms && nperm > 0)
        SQGR_HIP(hipMemcpyAsync(out_perms, p->perms_dev.p, (size_t)nperm * K2 * 4, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_nhood_shuffled_labels(sqgr_nhood* plan, uint64_t seed, int64_t perm, uint8_t* out_labels) {
    SQGR_REQUIRE(plan && out_labels && perm >= 0, "plan/out_labels is NULL or perm < 0");
    SQGR_REQUIRE(plan->K <= 256, "sqgr_nhood_shuffled_labels returns uint8 labels: K=%d > 256", plan->K);
    SQGR_REQUIRE(!plan->mapped(), "a plan with a spot map (sqgr_nhood_set_spot_map) runs sqgr_nhood_run only");
    sqgr_nhood* p = plan;
    sqgr_ctx* ctx = p->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    SQGR_TRY(p->ensure_workspace(false));
    hipStream_t st = ctx->stream;
    const int B = p->B;
    const int64_t perm_row = perm - perm % FEISTEL_GROUP;  // rows start at group boundaries of the global index
    {
        const char* e = getenv("SQGR_SHUFFLE_INDEPENDENT");
        p->independent = e && atoi(e) == 1;
        SQGR_REQUIRE(!p->independent || (B == 16 && !p->has_libs && !p->wide()), "SQGR_SHUFFLE_INDEPENDENT needs 16-wide rows, K <= 256 and no libraries");
    }
    if (p->independent)
        k_keygen_indep<<<1, 256, 0, st>>>(seed, perm_row, 1, p->keys.p);
    else
        k_keygen<<<(unsigned)ceil_div((int64_t)p->n_libs * (B / FEISTEL_GROUP + B / 2), 256), 256, 0, st>>>(seed, perm_row, 1, B, p->n_libs,
                                                                                                           p->keys.p);
    SQGR_HIP(hipGetLastError());
    SQGR_TRY(launch_shuffle(p, 1, 0, st));
    std::vector<uint8_t> rows((size_t)p->n * B);
    SQGR_HIP(hipMemcpyAsync(rows.data(), p->slab.p, rows.size(), hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    const int pw = p->plane_w(), slot = (int)(perm - perm_row);
    for (int64_t i = 0; i < p->n; ++i)
        out_labels[i] = pw == 16 || B != 16 ? rows[(size_t)i * B + (size_t)slot]
                                            : rows[(size_t)(slot / pw) * p->n * pw + (size_t)i * pw + (size_t)(slot % pw)];
    return SQGR_OK;
}

int sqgr_nhood_counts_batch(sqgr_ctx* ctx, const sqgr_graph* g, const uint8_t* labels, int64_t n_perms, int32_t K,
                            uint32_t* out_counts) {
    SQGR_REQUIRE(ctx && g && labels && out_counts, "ctx/graph/labels/out is NULL");
    SQGR_REQUIRE(n_perms >= 0, "n_perms < 0");
    SQGR_REQUIRE(K <= 256, "sqgr_nhood_counts_batch takes uint8 labels: K=%d > 256", K);
    const int64_t n = g->n;
    for (int64_t t = 0; t < n_perms * n; ++t)
        SQGR_REQUIRE(labels[t] < K, "labels[%lld]=%d outside [0,%d)", (long long)t, (int)labels[t], K);
    sqgr_nhood* p = nullptr;
    SQGR_TRY(sqgr_nhood_create(ctx, g, nullptr, K, nullptr, 0, &p));
    int rc = SQGR_OK;
    do {
        if ((rc = p->ensure_workspace(true)) != SQGR_OK) break;
        const int B = p->B, K2 = p->K2, hw = p->acc_words();
        const int64_t per_launch = (int64_t)p->nbatch * B;
        hipStream_t st = ctx->stream;
        if ((rc = p->stage.ensure((size_t)per_launch * n)) != SQGR_OK) break;
        if ((rc = p->perms_dev.ensure((size_t)(n_perms > 0 ? n_perms : 1) * K2)) != SQGR_OK) break;
        hipError_t e = hipMemsetAsync(p->shift.p, 0, (size_t)K2 * 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(p->acc_sum.p, 0, (size_t)p->nbatch * hw * 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(p->acc_sq.p, 0, (size_t)p->nbatch * hw * 8, st);
        for (int64_t p0 = 0; p0 < n_perms && e == hipSuccess && rc == SQGR_OK; p0 += per_launch) {
            const int64_t todo = (n_perms - p0 < per_launch) ? n_perms - p0 : per_launch;
            const int nb = (int)ceil_div(todo, B);
            e = hipMemcpyAsync(p->stage.p, labels + (size_t)p0 * n, (size_t)todo * n, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) break;
            {
                LaunchTimer t(ctx, "nhood_transpose_labels");
                if (B == 32)
                    k_transpose_labels<32><<<dim3((unsigned)ceil_div(n, 256), nb), 256, 0, st>>>(n, p->stage.p, todo, p->slab.p, 16);
                else
                    k_transpose_labels<16><<<dim3((unsigned)ceil_div(n, 256), nb), 256, 0, st>>>(n, p->stage.p, todo, p->slab.p, p->plane_w());
            }
            if ((rc = p->count_batches(nb, 0)) != SQGR_OK) break;
            if ((rc = p->reduce_batches(nb, p0, 0, n_perms, p->perms_dev.p)) != SQGR_OK) break;
            e = hipStreamSynchronize(st);  // stage buffer is reused by the next chunk
        }
        if (rc != SQGR_OK) break;
        if (e == hipSuccess && n_perms > 0)
            e = hipMemcpyAsync(out_counts, p->perms_dev.p, (size_t)n_perms * K2 * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error("nhood_counts_batch failed: %s", hipGetErrorString(e));
            rc = SQGR_ERR_HIP;
        }
    } while (0);
    sqgr_nhood_destroy(p);
    return rc;
}

// keep_perms: the per-permutation counts stay in plan->perms_dev (copied to out_perms when that is not NULL)
static int run_pcg64_impl(sqgr_nhood* plan, const uint64_t* pcg_states, int64_t n_perms, const int64_t* shift, int64_t* out_sum,
                          uint64_t* out_sumsq, uint32_t* out_perms, bool keep_perms) {
    SQGR_REQUIRE(plan && pcg_states && out_sum && out_sumsq && n_perms >= 0, "null argument or n_perms < 0");
    SQGR_REQUIRE(!plan->mapped(), "a plan with a spot map (sqgr_nhood_set_spot_map) runs sqgr_nhood_run only: numpy's streams permute positions");
    sqgr_nhood* p = plan;
    keep_perms = keep_perms || out_perms != nullptr;
    SQGR_REQUIRE(p->has_labels, "plan was created without labels");
    sqgr_ctx* ctx = p->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int K2 = p->K2;
    auto local = [&]() -> int {  // this rank's own part; its outcome is agreed on in front of the collective (see sqgr_nhood_run)
    SQGR_TRY(p->ensure_workspace(keep_perms));
    const int B = p->B, hw = p->acc_words();
    const int64_t n = p->n;
    if (shift)
        SQGR_HIP(hipMemcpyAsync(p->shift.p, shift, (size_t)K2 * 8, hipMemcpyHostToDevice, st));
    else
        SQGR_HIP(hipMemsetAsync(p->shift.p, 0, (size_t)K2 * 8, st));
    SQGR_HIP(hipMemsetAsync(p->acc_sum.p, 0, (size_t)p->nbatch * hw * 8, st));
    SQGR_HIP(hipMemsetAsync(p->acc_sq.p, 0, (size_t)p->nbatch * hw * 8, st));
    if (keep_perms) SQGR_TRY(p->perms_dev.ensure((size_t)(n_perms + 1) * K2));  // + 1: the all-gather of ragged rank chunks sends one padded row
    if (p->wide()) {
        // more than 256 clusters: `Generator.shuffle(x)` leaves x[perm] behind with perm = the same generator's
        // `permutation(n)` (one Fisher-Yates, the same draws), so the permutations are drawn on the device (4*n bytes each) and
        // the 16-bit labels gathered through them.  Per-library sub-shuffles (`_shuffle_group`) are not offered on this path.
        if (p->has_libs) {
            set_error("K=%d > 256 clusters with library_key: numpy-stream shuffles are not available on the device", p->K);
            return SQGR_ERR_UNSUPPORTED;
        }
        size_t free_b = 0, total_b = 0;
        SQGR_HIP(hipMemGetInfo(&free_b, &total_b));
        const int64_t per_launch = (int64_t)p->nbatch * B;
        int64_t chunk = (int64_t)std::min<size_t>(free_b / 4, (size_t)16 << 30) / std::max<int64_t>(4 * n, 1);
        chunk = std::max<int64_t>(per_launch, chunk / per_launch * per_launch);
        chunk = std::min<int64_t>(chunk, ceil_div(std::max<int64_t>(n_perms, 1), per_launch) * per_launch);
        SQGR_TRY(p->perm_idx.ensure((size_t)chunk * n));
        SQGR_TRY(p->pcg_states.ensure((size_t)chunk * 4));
        for (int64_t c0 = 0; c0 < n_perms; c0 += chunk) {
            const int64_t pc = std::min(chunk, n_perms - c0);
            SQGR_HIP(hipMemcpyAsync(p->pcg_states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
            SQGR_TRY(pcg_permutations_dev(ctx, p->pcg_ws, n, p->pcg_states.p, pc, p->perm_idx.p, st));
            for (int64_t q0 = 0; q0 < pc; q0 += per_launch) {
                const int64_t todo = std::min(per_launch, pc - q0);
                const int nb = (int)ceil_div(todo, B);
                {
                    LaunchTimer t(ctx, "nhood_gather_labels16");
                    k_gather_labels16<<<dim3((unsigned)ceil_div(n, 256), nb), 256, 0, st>>>(n, p->base16.p, p->perm_idx.p, q0, pc,
                                                                                           reinterpret_cast<uint16_t*>(p->slab.p));
                    SQGR_HIP(hipGetLastError());
                }
                SQGR_TRY(p->count_batches(nb, 0));
                SQGR_TRY(p->reduce_batches(nb, c0 + q0, 0, c0 + pc, keep_perms ? p->perms_dev.p : nullptr));
            }
        }
    } else {
        // permutations per chunk: one thread each; the column matrix takes n bytes per permutation (<= 25 % of free HBM)
        size_t free_b = 0, total_b = 0;
        SQGR_HIP(hipMemGetInfo(&free_b, &total_b));
        int64_t budget = (int64_t)std::min<size_t>(free_b / 4, (size_t)64 << 30);
        int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(budget / std::max<int64_t>(n, 1), 1 << 17));
        chunk = std::min<int64_t>(chunk, ceil_div(std::max<int64_t>(n_perms, 1), 64) * 64) / 64 * 64;
        const int64_t stride = chunk;  // multiple of 64 => 16-byte aligned slab gathers
        // without libraries the shuffled ROWS go straight into the slab (k_rows_to_slab); with libraries (position != spot) through
        // the column matrix as before
        const bool via_rows = !p->has_libs && pcg_rows_available();
        if (!via_rows) SQGR_TRY(p->wcol.ensure((size_t)n * stride));
        SQGR_TRY(p->pcg_states.ensure((size_t)chunk * 4));
        const int64_t per_launch = (int64_t)p->nbatch * B;
        for (int64_t c0 = 0; c0 < n_perms; c0 += chunk) {
            const int64_t pc = std::min(chunk, n_perms - c0);
            SQGR_HIP(hipMemcpyAsync(p->pcg_states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
            int64_t row_stride = 0;
            if (via_rows)
                SQGR_TRY(pcg_shuffle_rows(ctx, p->pcg_ws, n, p->n_libs, p->lib_off.p, p->base_pos.p, p->pcg_states.p, pc, st, "nhood_pcg64_shuffle",
                                          &row_stride));
            else
                SQGR_TRY(pcg_shuffle_labels(ctx, p->pcg_ws, n, p->n_libs, p->lib_off.p, p->base_pos.p, p->pcg_states.p, pc, stride, p->wcol.p, st,
                                            "nhood_pcg64_shuffle"));
            for (int64_t q0 = 0; q0 < pc; q0 += per_launch) {
                const int64_t todo = std::min(per_launch, pc - q0);
                const int nb = (int)ceil_div(todo, B);
                if (via_rows) {
                    LaunchTimer t(ctx, "nhood_rows_to_slab");
                    dim3 grid((unsigned)ceil_div(ceil_div(n, 4), 256), nb);
                    if (B == 32) k_rows_to_slab<32><<<grid, 256, 0, st>>>(n, row_stride, p->pcg_ws.rows.p, q0, pc, p->slab.p, 16);
                    else k_rows_to_slab<16><<<grid, 256, 0, st>>>(n, row_stride, p->pcg_ws.rows.p, q0, pc, p->slab.p, p->plane_w());
                    SQGR_HIP(hipGetLastError());
                } else {
                    LaunchTimer t(ctx, "nhood_columns_to_slab");
                    dim3 grid((unsigned)ceil_div(n, 256), nb);
    #define SQGR_C2S(BB, LIBS) k_columns_to_slab<BB, LIBS><<<grid, 256, 0, st>>>(n, stride, p->wcol.p, q0, p->lib_of.p, p->rank_of.p, p->lib_off.p, p->slab.p, p->plane_w())
                    if (B == 32) { if (p->has_libs) SQGR_C2S(32, true); else SQGR_C2S(32, false); }
                    else { if (p->has_libs) SQGR_C2S(16, true); else SQGR_C2S(16, false); }
    #undef SQGR_C2S
                    SQGR_HIP(hipGetLastError());
                }
                SQGR_TRY(p->count_batches(nb, 0));
                // columns past `pc` of the last batch hold stale data: reduce masks permutations >= n_perms
                SQGR_TRY(p->reduce_batches(nb, c0 + q0, 0, c0 + pc, keep_perms ? p->perms_dev.p : nullptr));
            }
        }
    }
    {
        LaunchTimer t(ctx, "nhood_finalize");
        SQGR_HIP(hipMemsetAsync(p->fin.p, 0, (size_t)2 * K2 * 8, st));
        k_finalize<<<(unsigned)ceil_div(hw, 256), 256, 0, st>>>(p->acc_sum.p, p->acc_sq.p, p->nbatch, hw, p->acc_pw(), K2, p->fin.p,
                                                               reinterpret_cast<uint64_t*>(p->fin.p + K2));
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
    };
    // (The swap records of the bucketed replay — up to 36 GB at 1e6 spots x 8192 permutations per pass — stay with the plan's
    // workspace until the plan goes: releasing them after every run was tried for ADVICE r4 and made a persistent plan re-allocate
    // 35 GB per run, more than the parked-buffer pool keeps; the driver then took seconds for some of those hipMallocs
    // (bench.py's numpy leg: 61 k -> 3.9 k permutations/s in one lease).  The front ends create and close their plan per call, so
    // a call gives the memory back either way; a caller that keeps a plan decides with sqgr_nhood_destroy / sqgr_ctx_trim.)
    SQGR_TRY(comm_agree(p->comm, local(), st));
    // multi-GPU: the ranks ran disjoint permutation ranges; one RCCL all-reduce of the 2*K*K exact integer moments on the
    // device, then every rank copies out the global sums
    SQGR_TRY(comm_allreduce_i64_dev(p->comm, p->fin.p, (size_t)2 * K2, false, st));
    SQGR_HIP(hipMemcpyAsync(out_sum, p->fin.p, (size_t)K2 * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipMemcpyAsync(out_sumsq, p->fin.p + K2, (size_t)K2 * 8, hipMemcpyDeviceToHost, st));
    if (out_perms && n_perms > 0)
        SQGR_HIP(hipMemcpyAsync(out_perms, p->perms_dev.p, (size_t)n_perms * K2 * 4, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_nhood_run_pcg64(sqgr_nhood* plan, const uint64_t* pcg_states, int64_t n_perms, const int64_t* shift, int64_t* out_sum,
                         uint64_t* out_sumsq, uint32_t* out_perms) {
    return run_pcg64_impl(plan, pcg_states, n_perms, shift, out_sum, out_sumsq, out_perms, false);
}

int sqgr_nhood_run_pcg64_stats(sqgr_nhood* plan, const uint64_t* pcg_states, int64_t n_perms, double* out_mean, double* out_std) {
    SQGR_REQUIRE(plan && pcg_states && out_mean && out_std && n_perms >= 1, "null argument or n_perms < 1");
    const int K2 = plan->K2;
    std::vector<int64_t> s1((size_t)K2);
    std::vector<uint64_t> s2((size_t)K2);
    // several ranks: `pcg_states` holds ALL n_perms generator states on every rank; this rank runs the contiguous chunk
    // it owns (chunks of base + 1 permutations on the first `rem` ranks, base on the others), the per-permutation counts
    // are all-gathered on the device (RCCL) and every rank reduces the full set in numpy's order: the result does not
    // depend on the number of ranks, bit for bit.
    const int world = comm_world(plan->comm), rank = comm_rank(plan->comm);
    const int64_t base = n_perms / world, rem = n_perms % world;
    const int64_t lo = rank * base + std::min<int64_t>(rank, rem), mine = base + (rank < rem ? 1 : 0);
    sqgr_comm* comm = plan->comm;
    plan->comm = nullptr;  // the moments of the slice are not what is wanted here: no all-reduce inside the run
    const int rc = run_pcg64_impl(plan, pcg_states + (size_t)lo * 4, mine, nullptr, s1.data(), s2.data(), nullptr, true);
    plan->comm = comm;
    sqgr_ctx* ctx = plan->ctx;
    hipStream_t st = ctx->stream;
    SQGR_TRY(comm_agree(comm, rc, st));  // a rank whose slice failed does not leave the others waiting in the collectives
    // numpy's order over ALL permutations without gathering them (the all-gather of round 3 moved P*K*K*4 bytes — 360 MB at config
    // 5): the two running sums are chains (k_numpy_chain).  Round r: rank r adds its slice to the running sum it holds, then
    // every rank takes rank r's array (an all-gather of K*K doubles, 7 KB at K = 30; RCCL has no cheaper primitive bound here and
    // the payload is latency-bound either way).  world rounds per chain, two chains.  One rank may cut its own slice into
    // segments (SQGR_NUMPY_STATS_SEGMENTS, tests): the same chain, the same bits.
    const uint32_t* perms = plan->perms_dev.p;
    DevBuf<double> d_acc, d_mean, d_recv;
    SQGR_TRY(d_acc.alloc((size_t)K2));
    SQGR_TRY(d_mean.alloc((size_t)K2));
    if (world > 1) SQGR_TRY(d_recv.alloc((size_t)world * K2));
    int segments = 1;
    if (const char* e = getenv("SQGR_NUMPY_STATS_SEGMENTS")) segments = std::max(1, atoi(e));
    const unsigned gridc = (unsigned)ceil_div(K2, 64);
    auto chain = [&](const double* mean_dev) -> int {
        SQGR_HIP(hipMemsetAsync(d_acc.p, 0, (size_t)K2 * 8, st));
        for (int r = 0; r < world; ++r) {
            if (r == rank) {
                LaunchTimer t(ctx, "nhood_numpy_mean_std");
                for (int sgm = 0; sgm < segments; ++sgm) {
                    const int64_t q0 = mine * sgm / segments, q1 = mine * (sgm + 1) / segments;
                    if (q1 > q0) k_numpy_chain<<<gridc, 64, 0, st>>>(perms + (size_t)q0 * K2, q1 - q0, K2, mean_dev, d_acc.p);
                }
                SQGR_HIP(hipGetLastError());
            }
            if (world > 1) {
                SQGR_TRY(comm_allgather_dev(comm, d_acc.p, d_recv.p, (size_t)K2 * 8, st));
                SQGR_HIP(hipMemcpyAsync(d_acc.p, d_recv.p + (size_t)r * K2, (size_t)K2 * 8, hipMemcpyDeviceToDevice, st));
            }
        }
        return SQGR_OK;
    };
    SQGR_TRY(chain(nullptr));
    k_numpy_chain_finish<<<gridc, 64, 0, st>>>(d_acc.p, K2, (double)n_perms, 0);
    SQGR_HIP(hipMemcpyAsync(d_mean.p, d_acc.p, (size_t)K2 * 8, hipMemcpyDeviceToDevice, st));
    SQGR_TRY(chain(d_mean.p));
    k_numpy_chain_finish<<<gridc, 64, 0, st>>>(d_acc.p, K2, (double)n_perms, 1);
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipMemcpyAsync(out_mean, d_mean.p, (size_t)K2 * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipMemcpyAsync(out_std, d_acc.p, (size_t)K2 * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_nhood_info(sqgr_nhood* plan, int64_t* out_info) {
    SQGR_REQUIRE(plan && out_info, "plan/out_info is NULL");
    sqgr_nhood* p = plan;
    SQGR_HIP(hipSetDevice(p->ctx->device));
    SQGR_TRY(p->resolve_tuning());
    const bool lds_path = p->g && p->lds_path();
    if (lds_path) SQGR_TRY(p->g->ensure_half());
    const bool half = lds_path && p->g->sym_state == 1;
    out_info[0] = p->B;
    out_info[1] = p->nbatch;
    out_info[2] = lds_path ? p->blocks_for(p->nbatch, p->has_labels) : 1;  // (what sqgr_nhood_run launches; the other entry points may cut finer)
    out_info[3] = p->g ? (half ? p->g->n_half + p->g->n_self : p->g->nnz) : 0;
    out_info[4] = half ? (p->g->n_self > 0 ? 2 : 1) : 0;
    out_info[5] = p->hist_words();
    out_info[6] = half ? p->g->n_self : 0;
    out_info[7] = FEISTEL_GROUP;
    out_info[8] = lds_path ? (p->B == 32 ? 32 : p->be()) : 0;   // permutations per pass over the edge list (0: device-scope counters)
    out_info[9] = p->split();
    out_info[10] = p->cm();
    out_info[11] = (int64_t)p->part_words() * 4;                // bytes of one chunk's partial histograms (all passes)
    return SQGR_OK;
}

int sqgr_nhood_set_spot_map(sqgr_nhood* plan, const int32_t* spot_of) {
    SQGR_REQUIRE(plan, "plan is NULL");
    if (!spot_of) {
        plan->spot_of.release();
        return SQGR_OK;
    }
    SQGR_REQUIRE(!plan->has_libs && !plan->wide(), "a spot map needs a plan without libraries and at most 256 clusters");
    for (int64_t i = 0; i < plan->n; ++i)
        SQGR_REQUIRE(spot_of[i] >= 0 && spot_of[i] < plan->n, "spot_of[%lld]=%d outside [0,%lld)", (long long)i, spot_of[i], (long long)plan->n);
    SQGR_HIP(hipSetDevice(plan->ctx->device));
    SQGR_TRY(plan->spot_of.alloc((size_t)plan->n));
    SQGR_HIP(hipMemcpy(plan->spot_of.p, spot_of, (size_t)plan->n * 4, hipMemcpyHostToDevice));
    return SQGR_OK;
}

int sqgr_nhood_set_comm(sqgr_nhood* plan, sqgr_comm* comm) {
    SQGR_REQUIRE(plan, "plan is NULL");
    plan->comm = comm;  // NULL detaches
    return SQGR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- label shuffler (sqgr_shuffle.h)
// The two label generators of this file without a graph behind them: other permutation tests over cluster labels
// (ligrec) draw their shuffled label vectors here, so a given (seed, permutation index) means the same arrangement
// of the label multiset in every sq.gr function.
namespace sqgr {

int label_shuffler_create(sqgr_ctx* ctx, int64_t n, const int32_t* labels, int K, LabelShuffler** out) {
    SQGR_REQUIRE(ctx && labels && out && n > 0, "ctx/labels/out is NULL or n <= 0");
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_nhood* plan = nullptr;
    SQGR_TRY(nhood_build(ctx, nullptr, n, labels, K, nullptr, 0, &plan));
    *out = reinterpret_cast<LabelShuffler*>(plan);
    return SQGR_OK;
}

bool label_shuffler_wide(const LabelShuffler* s) { return reinterpret_cast<const sqgr_nhood*>(s)->wide(); }

size_t label_shuffler_key_words16() { return (size_t)key_words_per_row(16, 1); }

int label_shuffler_philox16(LabelShuffler* s, uint64_t seed, int64_t perm0, int nb, uint32_t* keys_ws, uint16_t* slab16, hipStream_t st) {
    sqgr_nhood* p = reinterpret_cast<sqgr_nhood*>(s);
    if (!p->wide() || perm0 % FEISTEL_GROUP != 0) {
        set_error("label_shuffler_philox16: needs more than 256 labels and perm0 (%lld) a multiple of %d", (long long)perm0, FEISTEL_GROUP);
        return SQGR_ERR_INVALID;
    }
    {
        LaunchTimer t(p->ctx, "ligrec_keygen", st);
        const int64_t nk = (int64_t)nb * (16 / FEISTEL_GROUP + 16 / 2);
        k_keygen<<<(unsigned)ceil_div(nk, 256), 256, 0, st>>>(seed, perm0, nb, 16, 1, keys_ws);
        SQGR_HIP(hipGetLastError());
    }
    return launch_shuffle_raw(p, 16, nb, keys_ws, reinterpret_cast<uint8_t*>(slab16), st);
}

int label_shuffler_pcg64_16(LabelShuffler* s, const uint64_t* states_dev, int64_t pc, uint16_t* slab16, hipStream_t st) {
    sqgr_nhood* p = reinterpret_cast<sqgr_nhood*>(s);
    if (!p->wide() || !p->base16.p) {
        set_error("label_shuffler_pcg64_16: needs more than 256 labels");
        return SQGR_ERR_INVALID;
    }
    const int64_t n = p->n;
    SQGR_TRY(p->perm_idx.ensure((size_t)pc * n));
    SQGR_TRY(pcg_permutations_dev(p->ctx, p->pcg_ws, n, states_dev, pc, p->perm_idx.p, st));
    const int nb = (int)ceil_div(pc, 16);
    LaunchTimer t(p->ctx, "ligrec_gather_labels16", st);
    k_gather_labels16<<<dim3((unsigned)ceil_div(n, 256), nb), 256, 0, st>>>(n, p->base16.p, p->perm_idx.p, 0, pc, slab16);
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

void label_shuffler_destroy(LabelShuffler* s) { delete reinterpret_cast<sqgr_nhood*>(s); }

int label_shuffler_philox(LabelShuffler* s, uint64_t seed, int64_t perm0, int nb, uint32_t* keys_ws, uint8_t* slab,
                          hipStream_t st) {
    sqgr_nhood* p = reinterpret_cast<sqgr_nhood*>(s);
    {
        LaunchTimer t(p->ctx, "ligrec_keygen", st);
        if (perm0 % FEISTEL_GROUP != 0) {
            set_error("label_shuffler_philox: perm0=%lld is not a multiple of %d", (long long)perm0, FEISTEL_GROUP);
            return SQGR_ERR_INVALID;
        }
        const int64_t nk = (int64_t)nb * (32 / FEISTEL_GROUP + 32 / 2);
        k_keygen<<<(unsigned)ceil_div(nk, 256), 256, 0, st>>>(seed, perm0, nb, 32, 1, keys_ws);
        SQGR_HIP(hipGetLastError());
    }
    return launch_shuffle_raw(p, 32, nb, keys_ws, slab, st);
}

int label_shuffler_pcg64(LabelShuffler* s, const uint64_t* states_dev, int64_t pc, int64_t stride, uint8_t* W, hipStream_t st) {
    sqgr_nhood* p = reinterpret_cast<sqgr_nhood*>(s);
    return pcg_shuffle_labels(p->ctx, p->pcg_ws, p->n, 1, p->lib_off.p, p->base_pos.p, states_dev, pc, stride, W, st, "ligrec_pcg64_shuffle");
}

}  // namespace sqgr
