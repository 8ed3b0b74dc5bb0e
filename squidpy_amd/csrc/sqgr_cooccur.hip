// libsqgr: co-occurrence pair counting — the all-pairs radial scan of `_occur_count`
// (/root/reference/src/squidpy/gr/_ppatterns.py:283-310):
//     counts[a, b, r] = #{ i != j : lab_i = a, lab_j = b, d2_ij <= thr[r] },   d2 = dx*dx + dy*dy in float32.
//
// MI355X design.  The reference materialises an N x (L*K*K) int32 scratch (176 GB at
// 1e6 points); here points are counting-sorted by label on the host and cut into 256-point tiles, so a pair of
// tiles has ONE label pair (a, b) and the K*K dimension disappears from the inner loop:
//   * a workgroup owns tile ti (one point per thread, in registers) and sweeps a chunk of tiles tj >= ti; the
//     tj points are wave-uniform, so they arrive through scalar loads (SMEM), not LDS or VMEM;
//   * d2 is evaluated once per unordered pair (dx -> -dx leaves d2 bit-identical) and credited to (a,b) and (b,a);
//   * the L cumulative compares become one bin index: a conservative LDS lookup table over d2 gives a lower bound,
//     an exact compare loop against the float32 thresholds finishes it (same `d2 <= thr[r]` decisions, bit for bit);
//   * per-thread private histogram columns in LDS (hist[bin][thread]): ds_add_u32 with no bank conflicts and no
//     contention; flushed with 64-bit global atomics only when the label of the tj segment changes.
// Counts are integers => order independent => deterministic, and exact (uint64).
#include "sqgr_common.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace sqgr {

constexpr int CO_TILE = 256;
constexpr int CO_CELLS_MIN = 2048;    // d2 lookup cells (doubled up to CO_CELLS_MAX until <= 2 thresholds per 2 cells)
constexpr int CO_CELLS_MAX = 32768;
constexpr int CO_LMAX = 120;          // thresholds per sweep: (L+?)*1 KiB of private histogram columns must fit LDS
constexpr int CO_BATCH = 8;           // pairs in flight per thread in the branch-free kernel
constexpr int CO_CHUNK_TILES = 64;
constexpr int CO_TRASH = 3;           // overflow rows behind the L bins of the branch-free kernel (bin <= L + 2)
constexpr int CO_HC = 64;             // histogram columns of the branch-free kernel: one per LANE, shared by the block's four waves through
                                      // the LDS atomics they are anyway (round 3: 14 KB instead of 53 KB at 49 thresholds: 7 instead of 2 blocks per CU);
                                      // a column receives at most 4 waves x CO_CHUNK_TILES x 256 counts between two flushes

struct CoParams {
    float inv_cell;
    int ncells;
    int T, L, K;
    int shard_index, shard_count;
    int finite;               // all coordinates are finite: full foreign batches may skip the per-pair checks
    unsigned long long* out;  // [K][K][L] per-bin (non cumulative) ordered pair counts
    const int3* chunks;       // LIST sweeps: {tile ti, first, last + 1 entry of `cand`} per block
    const int32_t* cand;      // LIST sweeps: the candidate QUARTER tiles 4 * tj + s (tj >= ti) of every tile, ascending per tile
    const float4* box64;      // LIST sweeps: bounding boxes of the quarter tiles (64 points each) — a wavefront of ti is one of them
    float tmax;               // LIST sweeps: the largest threshold
};

template <bool FMA>
__device__ __forceinline__ float dist2(float xi, float yi, float xj, float yj) {
    const float dx = xi - xj, dy = yi - yj;
    if (FMA) return fmaf(dx, dx, dy * dy);
    return __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));  // numpy / un-contracted semantics: every op rounded
}

// Short radii (round 6).  With the points of a cluster ordered along a space-filling curve a 256-point tile covers a small
// patch, and two tiles whose bounding boxes lie farther apart than the largest threshold hold no pair that any bin counts: the
// float32 d2 of a pair is >= the same expression evaluated on the boxes' gaps (every operation of dist2 is monotone in |dx|, |dy|,
// and fl(xj - xi) >= fl(lo_j - hi_i) for xj >= lo_j, xi <= hi_i) — so `d2 <= thr` is false for all of them and skipping the
// tile pair changes no count, bit for bit.  box[t] = {xmin, xmax, ymin, ymax} over the tile's valid points.
template <bool FMA>
__device__ __forceinline__ bool co_tiles_in_range(const float4 bi, const float4 bj, float tmax) {
    const float gx = fmaxf(0.f, fmaxf(bi.x - bj.y, bj.x - bi.y));
    const float gy = fmaxf(0.f, fmaxf(bi.z - bj.w, bj.z - bi.w));
    return dist2<FMA>(gx, gy, 0.f, 0.f) <= tmax;
}

// one block per tile ti of this shard: how many QUARTER tiles u = 4 * tj + s (64 points each, tj >= ti) are in range of ti's box
// (pass 1), or their ascending list (pass 2).  The sweep then tests every wavefront of ti — a quarter tile itself — against the
// candidate's box once more: the granularity of the skip is 64 x 64 points.
template <bool FMA, bool FILL>
__global__ __launch_bounds__(CO_TILE) void k_co_candidates(const float4* __restrict__ box, const float4* __restrict__ box64, int T, float tmax,
                                                           int shard_index, int shard_count, int32_t* __restrict__ count,
                                                           const int64_t* __restrict__ offset, int32_t* __restrict__ cand) {
    __shared__ int s_wave[CO_TILE / 64];
    __shared__ int s_run;
    const int ti = blockIdx.x * shard_count + shard_index;
    if (ti >= T) return;
    const float4 bi = box[ti];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_run = 0;
    __syncthreads();
    for (int base = 4 * ti; base < 4 * T; base += CO_TILE) {
        const int tj = base + t;  // (a quarter-tile index)
        const bool hit = tj < 4 * T && co_tiles_in_range<FMA>(bi, box64[tj], tmax);
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = s_run;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (FILL && hit) cand[offset[blockIdx.x] + before + __popcll(m & ((1ull << lane) - 1ull))] = tj;
        __syncthreads();
        if (t == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (!FILL && t == 0) count[blockIdx.x] = s_run;
}

// sum every bin's HC histogram columns and credit (a,b) [and (b,a)]; leaves the histogram zeroed.
template <int HC>
__device__ void co_flush(uint32_t* hist, int L, int K, int a, int b, bool mirror, unsigned long long* out) {
    __syncthreads();
    const int t = threadIdx.x;
    for (int g = t; g < L; g += CO_TILE) {
        unsigned long long s = 0;
        uint32_t* row = hist + g * HC;
        for (int k = 0; k < HC; ++k) {
            const int kk = (k + t) & (HC - 1);  // rotate: lanes of a wave walk distinct banks
            s += row[kk];
            row[kk] = 0;
        }
        if (s) {
            atomicAdd(&out[((size_t)a * K + b) * L + g], s);
            if (mirror) atomicAdd(&out[((size_t)b * K + a) * L + g], s);
        }
    }
    __syncthreads();
}

// xs/ys: label-sorted, tile-padded coordinates; tile_label/tile_valid: per tile; thr: [L] ascending float32
// thresholds; cell: [CO_CELLS] lower bound of the bin index of any d2 falling into that d2-cell.
// (separate __restrict__ const pointers: lets the compiler keep the wave-uniform tj loads on the scalar unit)
template <bool FMA>
__global__ __launch_bounds__(CO_TILE) void k_cooccur(const float* __restrict__ xs, const float* __restrict__ ys,
                                                     const int32_t* __restrict__ tile_label,
                                                     const int32_t* __restrict__ tile_valid, const float* __restrict__ thr,
                                                     const uint16_t* __restrict__ cell, CoParams p) {
    extern __shared__ uint32_t smem[];
    const int L = p.L;
    uint32_t* hist = smem;                                          // [L][256]
    float* s_thr = reinterpret_cast<float*>(smem + L * CO_TILE);    // [L]
    uint16_t* s_cell = reinterpret_cast<uint16_t*>(s_thr + L + 2);  // [ncells]
    const int t = threadIdx.x;

    const int ti = blockIdx.x * p.shard_count + p.shard_index;
    if (ti >= p.T) return;
    const int tj0 = max(ti, (int)blockIdx.y * CO_CHUNK_TILES);
    const int tj1 = min(p.T, ((int)blockIdx.y + 1) * CO_CHUNK_TILES);
    if (tj0 >= tj1) return;

    for (int i = t; i < L * CO_TILE; i += CO_TILE) hist[i] = 0;
    for (int i = t; i < L + 2; i += CO_TILE) s_thr[i] = i < L ? thr[i] : __builtin_inff();  // +inf sentinels
    for (int i = t; i < p.ncells; i += CO_TILE) s_cell[i] = cell[i];
    __syncthreads();

    const int a = tile_label[ti];
    const bool active = t < tile_valid[ti];
    const float xi = xs[(size_t)ti * CO_TILE + t];
    const float yi = ys[(size_t)ti * CO_TILE + t];
    const float inv_cell = p.inv_cell;
    uint32_t* my = hist + t;

    int cur_b = -1;
    for (int tj = tj0; tj < tj1; ++tj) {
        const int b = tile_label[tj];
        if (b != cur_b) {
            if (cur_b >= 0) co_flush<CO_TILE>(hist, L, p.K, a, cur_b, true, p.out);
            cur_b = b;
        }
        const int vj = tile_valid[tj];
        const float* __restrict__ xj = xs + (size_t)tj * CO_TILE;  // wave-uniform addresses: scalar loads
        const float* __restrict__ yj = ys + (size_t)tj * CO_TILE;
        const bool diag = (tj == ti);
        if (active) {
#pragma unroll 8
            for (int j = 0; j < vj; ++j) {
                const float d2 = dist2<FMA>(xi, yi, xj[j], yj[j]);
                int cellv = (int)(d2 * inv_cell);  // v_cvt_i32_f32 saturates; NaN -> 0
                cellv = min(max(cellv, 0), p.ncells - 1);
                int g = s_cell[cellv];
                while (g < L && !(d2 <= s_thr[g])) ++g;  // exact: first threshold with d2 <= thr (NaN never counts)
                if (g < L && !(diag && j == t)) atomicAdd(my + g * CO_TILE, 1u);
            }
        }
        if (diag) {  // ordered pairs of the diagonal tile are complete on their own: credit (a,a) once
            co_flush<CO_TILE>(hist, L, p.K, a, a, false, p.out);
        }
    }
    co_flush<CO_TILE>(hist, L, p.K, a, cur_b, true, p.out);
}

// Branch-free variant used whenever the lookup table is fine enough that the true bin is at most 2 above the
// table's lower bound (checked on the host): CO_BATCH pairs per thread are in flight, so the LDS round trips
// (cell lookup, two thresholds via one ds_read2, histogram add) overlap instead of serialising.
// LIST (round 6, short radii): the block does not sweep a range of tiles tj but one chunk of tile ti's CANDIDATE list — the tiles
// whose bounding box lies within the largest threshold of ti's (k_co_candidates) — named by p.chunks[blockIdx.x] = {ti, first, last + 1}.
template <bool FMA, bool LIST = false>
__global__ __launch_bounds__(CO_TILE) void k_cooccur_fast(const float* __restrict__ xs, const float* __restrict__ ys,
                                                          const int32_t* __restrict__ tile_label,
                                                          const int32_t* __restrict__ tile_valid, const float* __restrict__ thr,
                                                          const uint16_t* __restrict__ cell, CoParams p) {
    extern __shared__ uint32_t smem[];
    const int L = p.L;
    uint32_t* hist = smem;                                          // [L + CO_TRASH][CO_HC]: bins, then write-only overflow rows
    // [L + 2][32]: every threshold (and two +inf sentinels) 32 times, copy c in bank c — lane l reads copy l & 31, so the 64 lanes'
    // threshold reads (random g) never meet in a bank (tools/ubench_ds_mix.hip: the plain [L] array cost ~8 clk per ds_read2_b32)
    float* s_thr = reinterpret_cast<float*>(smem + (L + CO_TRASH) * CO_HC);
    uint16_t* s_cell = reinterpret_cast<uint16_t*>(s_thr + (L + 2) * 32);  // [ncells]
    const int t = threadIdx.x;

    int ti, tj0, tj1;  // LIST: tj0 / tj1 index the candidate list
    if constexpr (LIST) {
        const int3 ch = p.chunks[blockIdx.x];
        ti = ch.x;
        tj0 = ch.y;
        tj1 = ch.z;
    } else {
        ti = blockIdx.x * p.shard_count + p.shard_index;
        if (ti >= p.T) return;
        tj0 = max(ti, (int)blockIdx.y * CO_CHUNK_TILES);
        tj1 = min(p.T, ((int)blockIdx.y + 1) * CO_CHUNK_TILES);
        if (tj0 >= tj1) return;
    }

    for (int i = t; i < (L + CO_TRASH) * CO_HC; i += CO_TILE) hist[i] = 0;
    for (int i = t; i < (L + 2) * 32; i += CO_TILE) s_thr[i] = (i >> 5) < L ? thr[i >> 5] : __builtin_inff();
    for (int i = t; i < p.ncells; i += CO_TILE) s_cell[i] = cell[i];
    __syncthreads();

    const int a = tile_label[ti];
    const bool active = t < tile_valid[ti];
    const float xi = xs[(size_t)ti * CO_TILE + t];
    const float yi = ys[(size_t)ti * CO_TILE + t];
    const float inv_cell = p.inv_cell;
    const int cmax = p.ncells - 1;
    uint32_t* my = hist + (t & (CO_HC - 1));
    const int l32 = t & 31;

    float4 wbox = make_float4(0.f, 0.f, 0.f, 0.f);  // LIST: the box of this wavefront's 64 points of ti
    if constexpr (LIST) wbox = p.box64[4 * ti + (t >> 6)];
    int cur_b = -1;
    for (int it = tj0; it < tj1; ++it) {
        const int ent = LIST ? p.cand[it] : it;
        const int tj = LIST ? ent >> 2 : ent;
        const int b = tile_label[tj];
        if (b != cur_b) {
            if (cur_b >= 0) co_flush<CO_HC>(hist, L, p.K, a, cur_b, true, p.out);
            cur_b = b;
        }
        // LIST: points [jlo, vj) of tile tj = one quarter tile; the wavefront skips it when ITS box is out of range too (wave-uniform)
        const int jlo = LIST ? (ent & 3) * 64 : 0;
        const int vj = LIST ? min(tile_valid[tj], jlo + 64) : tile_valid[tj];
        bool wave_on = true;  // (the skip stays inside the wavefront: the flushes below are block-wide barriers)
        if constexpr (LIST) wave_on = co_tiles_in_range<FMA>(wbox, p.box64[ent], p.tmax);
        const float* __restrict__ xj = xs + (size_t)tj * CO_TILE;  // wave-uniform addresses: scalar loads; the tile is
        const float* __restrict__ yj = ys + (size_t)tj * CO_TILE;  // zero-padded to 256 so reading past vj is safe
        const int self = (tj == ti) ? t : -1;
        // One batch of CO_BATCH points of tile tj against this thread's point.  CHECKED: the general form (tail batch
        // of a tile, diagonal tile: j < vj, j != self, NaN, bin < L tested and folded into the increment).  Unchecked:
        // a full batch of a foreign tile with finite coordinates needs none of it — a pair beyond the last threshold
        // lands in one of the CO_TRASH overflow rows (bin <= L + 2 by construction of the cell table) that are never read.
        auto batch = [&](int j0, auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
            float d2[CO_BATCH];
            int g[CO_BATCH];
#pragma unroll
            for (int u = 0; u < CO_BATCH; ++u) {
                d2[u] = dist2<FMA>(xi, yi, xj[j0 + u], yj[j0 + u]);
                int cellv = (int)(d2[u] * inv_cell);  // v_cvt_i32_f32 saturates; NaN -> 0
                if constexpr (CHECKED) cellv = max(cellv, 0);  // finite d2 >= 0 cannot go negative
                g[u] = s_cell[min(cellv, cmax)];
            }
            // all threshold pairs are fetched before the first histogram update: the compiler may not move an LDS read
            // across an LDS atomic, so one loop would pay the LDS latency once per pair instead of once per batch
            float t0[CO_BATCH], t1[CO_BATCH];
#pragma unroll
            for (int u = 0; u < CO_BATCH; ++u) {
                t0[u] = s_thr[g[u] * 32 + l32];  // 32 words apart: one ds_read2_b32
                t1[u] = s_thr[g[u] * 32 + 32 + l32];
            }
#pragma unroll
            for (int u = 0; u < CO_BATCH; ++u) {
                // thresholds ascend, so !(d2 <= t1) implies !(d2 <= t0): bin = g + c0 + c1 (no branches)
                const int c0 = !(d2[u] <= t0[u]), c1 = !(d2[u] <= t1[u]);
                const int gg = g[u] + c0 + c1;
                if constexpr (CHECKED) {
                    const int ok = (int)(gg < L) & (int)(j0 + u < vj) & (int)(j0 + u != self) & (int)(d2[u] == d2[u]);
                    atomicAdd(my + min(gg, L - 1) * CO_HC, (uint32_t)ok);
                } else {
                    atomicAdd(my + gg * CO_HC, 1u);
                }
            }
        };
        if (active && wave_on) {
            if (tj == ti || !p.finite) {
                for (int j0 = jlo; j0 < vj; j0 += CO_BATCH) batch(j0, std::true_type{});
            } else {
                const int jfull = jlo + ((vj - jlo) & ~(CO_BATCH - 1));
                for (int j0 = jlo; j0 < jfull; j0 += CO_BATCH) batch(j0, std::false_type{});
                if (jfull < vj) batch(jfull, std::true_type{});
            }
        }
        // diagonal tile: ordered pairs complete, credit (a,a) once (LIST: behind the last of its quarter tiles — they open ti's list, in
        // the first chunk)
        if (tj == ti && (!LIST || it + 1 == tj1 || (p.cand[it + 1] >> 2) != ti)) co_flush<CO_HC>(hist, L, p.K, a, a, false, p.out);
    }
    co_flush<CO_HC>(hist, L, p.K, a, cur_b, true, p.out);
}

}  // namespace sqgr

using namespace sqgr;

extern "C" int sqgr_cooccur_counts(sqgr_ctx* ctx, const float* x, const float* y, const int32_t* labels, int64_t n,
                                   int32_t K, const float* thr2, int32_t L, int32_t fma, int32_t shard_index,
                                   int32_t shard_count, int64_t* out_counts) {
    SQGR_REQUIRE(ctx && x && y && labels && thr2 && out_counts, "null argument");
    SQGR_REQUIRE(n >= 0 && K >= 1 && L >= 1, "bad sizes n=%lld K=%d L=%d", (long long)n, K, L);
    SQGR_REQUIRE(shard_count >= 1 && shard_index >= 0 && shard_index < shard_count, "bad shard %d/%d", shard_index, shard_count);
    SQGR_REQUIRE(L <= 65535, "L=%d thresholds: too many", L);
    std::fill(out_counts, out_counts + (size_t)K * K * L, (int64_t)0);
    if (n == 0) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));

    // ---- counting sort by label, segments padded to whole tiles
    std::vector<int64_t> cnt((size_t)K, 0);
    for (int64_t i = 0; i < n; ++i) {
        SQGR_REQUIRE(labels[i] >= 0 && labels[i] < K, "labels[%lld]=%d outside [0,%d)", (long long)i, labels[i], K);
        cnt[labels[i]]++;
    }
    std::vector<int64_t> tile0((size_t)K + 1, 0);
    for (int k = 0; k < K; ++k) tile0[k + 1] = tile0[k] + ceil_div(cnt[k], CO_TILE);
    const int64_t T = tile0[K];
    SQGR_REQUIRE(T < (int64_t)1 << 30, "too many tiles");
    bool all_finite = true;
    float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        all_finite = all_finite && std::isfinite(x[i]) && std::isfinite(y[i]);
        xlo = std::min(xlo, x[i]); xhi = std::max(xhi, x[i]);
        ylo = std::min(ylo, y[i]); yhi = std::max(yhi, y[i]);
    }
    // Short radii: when the largest threshold is small against the extent of the cloud most tile pairs hold no countable pair.
    // The points of a cluster are then ordered along a Hilbert curve (counting sort by a 256 x 256 grid cell: O(n)), tiles become
    // compact patches, and only tile pairs whose bounding boxes lie within the largest threshold are swept (k_co_candidates) —
    // exact (see co_tiles_in_range).  SQGR_COOCCUR_SPARSE=0 | 1 switches the route off | on whatever the radius.
    float tmax_all = 0.f;
    bool thr_finite = true;
    for (int r = 0; r < L; ++r) {
        if (std::isnan(thr2[r])) continue;
        thr_finite = thr_finite && std::isfinite(thr2[r]);
        tmax_all = std::max(tmax_all, thr2[r]);
    }
    const float extent = std::max(xhi - xlo, yhi - ylo);
    bool sparse = all_finite && thr_finite && tmax_all > 0.f && extent > 0.f && T >= 64 &&
                  std::sqrt((double)tmax_all) < 0.2 * (double)extent;
    if (const char* e = getenv("SQGR_COOCCUR_SPARSE")) sparse = all_finite && thr_finite && tmax_all > 0.f && extent > 0.f && atoi(e) != 0;
    std::vector<float> xs((size_t)T * CO_TILE, 0.f), ys((size_t)T * CO_TILE, 0.f);
    std::vector<int32_t> tile_label((size_t)T), tile_valid((size_t)T);
    if (sparse) {
        // Two counting sorts, both over tables that stay in cache: by cluster (as the dense route), then inside every cluster by
        // the Hilbert index of the point's cell in a 256 x 256 grid over the cloud (a 64 K-entry histogram per cluster; the index
        // comes out of a table built once per process).  Consecutive cells of a Hilbert curve are always neighbours — a Z-order
        // curve jumps across the grid at every quadrant boundary, and a 64-point run that straddles one has a bounding box as
        // large as the quadrant: measured 13.2 ms against 7.3 ms for the sweep at 1e6 points, 30 clusters, radius 2 % of the extent.
        constexpr int GB = 8, G = 1 << GB;
        static const std::vector<uint16_t> hilbert_of = [] {
            std::vector<uint16_t> tab((size_t)G * G);
            for (uint32_t y0 = 0; y0 < (uint32_t)G; ++y0)
                for (uint32_t x0 = 0; x0 < (uint32_t)G; ++x0) {
                    uint32_t cx = x0, cy = y0, d = 0;
                    for (uint32_t sft = 1u << (GB - 1); sft > 0; sft >>= 1) {
                        const uint32_t rx = (cx & sft) ? 1u : 0u, ry = (cy & sft) ? 1u : 0u;
                        d += sft * sft * ((3u * rx) ^ ry);
                        if (ry == 0) {
                            if (rx == 1) {
                                cx = sft - 1 - (cx & (sft - 1));  // (only the low bits matter from here on)
                                cy = sft - 1 - (cy & (sft - 1));
                            }
                            const uint32_t tmp = cx;
                            cx = cy;
                            cy = tmp;
                        }
                        cx &= sft - 1;
                        cy &= sft - 1;
                    }
                    tab[(size_t)y0 * G + x0] = (uint16_t)d;
                }
            return tab;
        }();
        const float sx = (float)((double)G / ((double)xhi - (double)xlo + 1e-30)), sy = (float)((double)G / ((double)yhi - (double)ylo + 1e-30));
        // pass 1: by cluster, in input order (sequential writes per cluster), the cell's Hilbert index beside the coordinates
        std::vector<float> tx((size_t)n), ty((size_t)n);
        std::vector<uint16_t> tcode((size_t)n);
        std::vector<int64_t> first((size_t)K + 1, 0), fill((size_t)K, 0);
        for (int k = 0; k < K; ++k) first[k + 1] = first[k] + cnt[k];
        for (int64_t i = 0; i < n; ++i) {
            const int k = labels[i];
            const int64_t pos = first[k] + fill[k]++;
            const int cx = std::min(G - 1, std::max(0, (int)((x[i] - xlo) * sx))), cy = std::min(G - 1, std::max(0, (int)((y[i] - ylo) * sy)));
            tx[pos] = x[i];
            ty[pos] = y[i];
            tcode[pos] = hilbert_of[(size_t)cy * G + cx];
        }
        // pass 2: inside every cluster by Hilbert index (any order inside a cell)
        // (a coarser level of the curve for small clusters — the leading bits of a Hilbert index are the index one level up —: about
        // two cells per point, so that the histogram never costs more than the cluster it sorts)
        std::vector<uint32_t> hist((size_t)G * G + 1);
        for (int k = 0; k < K; ++k) {
            int lv = 1;
            while (lv < GB && ((int64_t)1 << (2 * lv)) < 2 * cnt[k]) ++lv;
            const int drop = 2 * (GB - lv);
            const size_t cells = (size_t)1 << (2 * lv);
            std::fill(hist.begin(), hist.begin() + cells + 1, 0u);
            for (int64_t q = first[k]; q < first[k + 1]; ++q) hist[(size_t)(tcode[q] >> drop) + 1]++;
            for (size_t c = 0; c < cells; ++c) hist[c + 1] += hist[c];
            const int64_t base = tile0[k] * CO_TILE;
            for (int64_t q = first[k]; q < first[k + 1]; ++q) {
                const int64_t pos = base + hist[tcode[q] >> drop]++;
                xs[pos] = tx[q];
                ys[pos] = ty[q];
            }
        }
    } else {
        std::vector<int64_t> fill((size_t)K, 0);
        for (int64_t i = 0; i < n; ++i) {
            const int k = labels[i];
            const int64_t pos = tile0[k] * CO_TILE + fill[k]++;
            xs[pos] = x[i];
            ys[pos] = y[i];
        }
    }
    for (int k = 0; k < K; ++k)
        for (int64_t tt = tile0[k]; tt < tile0[k + 1]; ++tt) {
            tile_label[tt] = k;
            const int64_t left = cnt[k] - (tt - tile0[k]) * CO_TILE;
            tile_valid[tt] = (int32_t)std::min<int64_t>(left, CO_TILE);
        }
    // candidate tiles of the short-radius route: boxes of the valid points, lists built on the device once for all threshold chunks
    DevBuf<float4> d_box, d_box64;
    DevBuf<int32_t> d_ccount, d_cand;
    DevBuf<int64_t> d_coff;
    DevBuf<int3> d_chunks;
    int64_t n_chunks = 0, n_cand = 0;
    const int64_t Ts = ceil_div(T, shard_count);  // tiles ti of this shard (blockIdx.x -> ti = blockIdx.x * shard_count + shard_index)
    if (sparse && T >= ((int64_t)1 << 28)) sparse = false;  // (quarter-tile indices are int32)
    if (sparse) {
        std::vector<float4> box((size_t)T), box64((size_t)T * 4);
        for (int64_t tt = 0; tt < T; ++tt) {
            float4 b = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
            for (int q = 0; q < 4; ++q) {
                float4 bq = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);  // an empty quarter keeps this box: in range of nothing
                for (int j = q * 64; j < std::min(tile_valid[tt], (q + 1) * 64); ++j) {
                    const float px = xs[(size_t)tt * CO_TILE + j], py = ys[(size_t)tt * CO_TILE + j];
                    bq.x = std::min(bq.x, px); bq.y = std::max(bq.y, px);
                    bq.z = std::min(bq.z, py); bq.w = std::max(bq.w, py);
                }
                box64[(size_t)tt * 4 + q] = bq;
                b.x = std::min(b.x, bq.x); b.y = std::max(b.y, bq.y);
                b.z = std::min(b.z, bq.z); b.w = std::max(b.w, bq.w);
            }
            box[tt] = b;
        }
        hipStream_t st = ctx->stream;
        SQGR_TRY(d_box.alloc((size_t)T));
        SQGR_TRY(d_box64.alloc((size_t)T * 4));
        SQGR_HIP(hipMemcpyAsync(d_box64.p, box64.data(), (size_t)T * 4 * sizeof(float4), hipMemcpyHostToDevice, st));
        SQGR_TRY(d_ccount.alloc((size_t)Ts));
        SQGR_TRY(d_coff.alloc((size_t)Ts + 1));
        SQGR_HIP(hipMemcpyAsync(d_box.p, box.data(), (size_t)T * sizeof(float4), hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemsetAsync(d_ccount.p, 0, (size_t)Ts * 4, st));
        {
            LaunchTimer tm(ctx, "cooccur_candidates");
            if (fma) k_co_candidates<true, false><<<(unsigned)Ts, CO_TILE, 0, st>>>(d_box.p, d_box64.p, (int)T, tmax_all, shard_index, shard_count, d_ccount.p, nullptr, nullptr);
            else k_co_candidates<false, false><<<(unsigned)Ts, CO_TILE, 0, st>>>(d_box.p, d_box64.p, (int)T, tmax_all, shard_index, shard_count, d_ccount.p, nullptr, nullptr);
            SQGR_HIP(hipGetLastError());
        }
        std::vector<int32_t> ccount((size_t)Ts);
        SQGR_HIP(hipMemcpyAsync(ccount.data(), d_ccount.p, (size_t)Ts * 4, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
        std::vector<int64_t> coff((size_t)Ts + 1, 0);
        std::vector<int3> chunks;
        for (int64_t b = 0; b < Ts; ++b) {
            coff[b + 1] = coff[b] + ccount[b];
            const int64_t ti = b * shard_count + shard_index;
            if (ti >= T) continue;
            for (int64_t c0 = coff[b]; c0 < coff[b + 1]; c0 += 4 * CO_CHUNK_TILES)  // (quarter tiles: the same work per block as the dense sweep)
                chunks.push_back(make_int3((int)ti, (int)c0, (int)std::min<int64_t>(coff[b + 1], c0 + 4 * CO_CHUNK_TILES)));
        }
        n_cand = coff[Ts];
        n_chunks = (int64_t)chunks.size();
        // not worth it when most tile pairs stay (a radius that reaches across the tiles): the dense sweep has no lists to walk
        const double dense_pairs = 0.5 * 4.0 * (double)T * (double)T / (double)shard_count;  // (in quarter tiles)
        const int64_t cand_cap = (int64_t)500 << 20;  // 2 GB of candidate entries at most (the dense sweep needs no list at all)
        if (n_cand >= (int64_t)(0.5 * dense_pairs) || n_cand >= cand_cap || n_chunks == 0) {
            if (!getenv("SQGR_COOCCUR_SPARSE") || n_chunks == 0 || n_cand >= cand_cap) sparse = false;
        }
        if (const char* e = getenv("SQGR_COOCCUR_DEBUG"))
            if (atoi(e))
                fprintf(stderr, "sqgr co_occurrence: %lld tiles, %lld candidate quarter tiles (dense: %.3g), %lld chunks, route %s\n", (long long)T, (long long)n_cand,
                        dense_pairs, (long long)n_chunks, sparse ? "near" : "dense");
        if (sparse) {
            SQGR_TRY(d_cand.alloc((size_t)std::max<int64_t>(n_cand, 1)));
            SQGR_TRY(d_chunks.alloc((size_t)n_chunks));
            SQGR_HIP(hipMemcpyAsync(d_coff.p, coff.data(), coff.size() * 8, hipMemcpyHostToDevice, st));
            SQGR_HIP(hipMemcpyAsync(d_chunks.p, chunks.data(), chunks.size() * sizeof(int3), hipMemcpyHostToDevice, st));
            LaunchTimer tm(ctx, "cooccur_candidates");
            if (fma) k_co_candidates<true, true><<<(unsigned)Ts, CO_TILE, 0, st>>>(d_box.p, d_box64.p, (int)T, tmax_all, shard_index, shard_count, nullptr, d_coff.p, d_cand.p);
            else k_co_candidates<false, true><<<(unsigned)Ts, CO_TILE, 0, st>>>(d_box.p, d_box64.p, (int)T, tmax_all, shard_index, shard_count, nullptr, d_coff.p, d_cand.p);
            SQGR_HIP(hipGetLastError());
            SQGR_HIP(hipStreamSynchronize(st));  // (`coff`, `chunks` leave scope with this block)
        }
    }

    // ---- thresholds: ascending order (the count for a threshold depends on its value only)
    std::vector<int> order((size_t)L);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) {
        const float a = thr2[p], b = thr2[q];
        if (std::isnan(a)) return false;  // NaNs last: `d2 <= NaN` is never true
        if (std::isnan(b)) return true;
        return a < b;
    });
    std::vector<float> thr_s((size_t)L);
    for (int r = 0; r < L; ++r) thr_s[r] = thr2[order[r]];
    int L_eff = L;
    while (L_eff > 0 && std::isnan(thr_s[L_eff - 1])) --L_eff;  // NaN thresholds count nothing
    // cum_host[ab][g] = #{pairs of label pair ab with d2 <= thr_s[g]}.  Thresholds are swept in chunks of <= CO_LMAX
    // bins (LDS capacity); a chunk's cumulative histogram is already the full cumulative count for its thresholds,
    // because every d2 below the chunk's first threshold lands in the chunk's bin 0.
    std::vector<unsigned long long> cum_host((size_t)K * K * L, 0ull);
    const int L_all = L_eff;
    const float* thr_all = thr_s.data();
    for (int s0 = 0; s0 < L_all; s0 += CO_LMAX) {
        const int L_eff = std::min(CO_LMAX, L_all - s0);
        const float* thr_s = thr_all + s0;
        float tmax = 0.f;  // the lookup table spans [0, largest finite threshold]; +inf thresholds live past its last cell
        for (int g = 0; g < L_eff; ++g)
            if (std::isfinite(thr_s[g])) tmax = std::max(tmax, thr_s[g]);
        const bool table_ok = tmax > 0.f;
        const size_t lds_fixed = (size_t)(L_eff + CO_TRASH) * CO_TILE * 4 + (size_t)(L_eff + 2) * 4;
        int ncells = CO_CELLS_MIN;
        float inv_cell = 0.f;
        std::vector<uint16_t> cell;
        bool fast = false;
        for (;; ncells *= 2) {
            inv_cell = table_ok ? (float)((double)ncells / (double)tmax) : 0.f;
            cell.assign((size_t)ncells, 0);
            int worst = L_eff;  // max (true bin - table lower bound) over all cells
            if (inv_cell > 0.f && std::isfinite(inv_cell)) {
                worst = 0;
                for (int c = 0; c < ncells; ++c) {
                    // every d2 landing in cell c lies in [(c-1)/inv_cell, (c+2)/inv_cell): a full cell of slack on both
                    // sides absorbs the float rounding of d2 * inv_cell
                    const double lo = (c >= 1) ? ((double)(c - 1) / (double)inv_cell) * (1.0 - 1e-5) : -1.0;
                    const double hi = (c == ncells - 1) ? (double)INFINITY : ((double)(c + 2) / (double)inv_cell) * (1.0 + 1e-5);
                    int g = 0;
                    while (g < L_eff && (double)thr_s[g] < lo) ++g;
                    int gh = g;
                    while (gh < L_eff && (double)thr_s[gh] < hi) ++gh;
                    cell[c] = (uint16_t)g;
                    worst = std::max(worst, gh - g);
                }
            }
            fast = worst <= 2;
            if (fast || ncells * 2 > CO_CELLS_MAX || lds_fixed + (size_t)ncells * 4 > 160 * 1024) break;
        }
        if (!fast && lds_fixed + (size_t)ncells * 2 > 160 * 1024) {  // fall back to the smallest table
            ncells = CO_CELLS_MIN;
            inv_cell = table_ok ? (float)((double)ncells / (double)tmax) : 0.f;
            cell.assign((size_t)ncells, 0);
            if (inv_cell > 0.f && std::isfinite(inv_cell))
                for (int c = 0; c < ncells; ++c) {
                    const double lo = (c >= 1) ? ((double)(c - 1) / (double)inv_cell) * (1.0 - 1e-5) : -1.0;
                    int g = 0;
                    while (g < L_eff && (double)thr_s[g] < lo) ++g;
                    cell[c] = (uint16_t)g;
                }
        }
        DevBuf<float> d_x, d_y, d_thr;
        DevBuf<int32_t> d_tl, d_tv;
        DevBuf<uint16_t> d_cell;
        DevBuf<unsigned long long> d_out;
        SQGR_TRY(d_x.alloc(xs.size()));
        SQGR_TRY(d_y.alloc(ys.size()));
        SQGR_TRY(d_thr.alloc((size_t)L_eff));
        SQGR_TRY(d_tl.alloc((size_t)T));
        SQGR_TRY(d_tv.alloc((size_t)T));
        SQGR_TRY(d_cell.alloc((size_t)ncells));
        SQGR_TRY(d_out.alloc((size_t)K * K * L_eff));
        hipStream_t st = ctx->stream;
        SQGR_HIP(hipMemcpyAsync(d_x.p, xs.data(), xs.size() * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(d_y.p, ys.data(), ys.size() * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(d_thr.p, thr_s, (size_t)L_eff * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(d_tl.p, tile_label.data(), (size_t)T * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(d_tv.p, tile_valid.data(), (size_t)T * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(d_cell.p, cell.data(), (size_t)ncells * 2, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemsetAsync(d_out.p, 0, (size_t)K * K * L_eff * 8, st));
        CoParams p{inv_cell, ncells, (int)T, L_eff, K, shard_index, shard_count, all_finite ? 1 : 0, d_out.p, d_chunks.p, d_cand.p, d_box64.p, tmax_all};
        const size_t lds_eff = fast ? (size_t)(L_eff + CO_TRASH) * CO_HC * 4 + (size_t)(L_eff + 2) * 32 * 4 + (size_t)ncells * 2 : lds_fixed + (size_t)ncells * 2;
        dim3 grid((unsigned)ceil_div(T, shard_count), (unsigned)ceil_div(T, CO_CHUNK_TILES));
        const bool list = sparse && fast;  // (the exact-compare kernel of coarse tables keeps the dense sweep: same counts)
        if (list) grid = dim3((unsigned)n_chunks, 1);
        {
            LaunchTimer tm(ctx, list ? (fma ? "cooccur_pairs_near_fma" : "cooccur_pairs_near")
                                     : (fast ? (fma ? "cooccur_pairs_fast_fma" : "cooccur_pairs_fast") : (fma ? "cooccur_pairs_fma" : "cooccur_pairs")));
#define SQGR_CO(KERNEL)                                                                                                          \
    do {                                                                                                                         \
        if (lds_eff > 64 * 1024)                                                                                                 \
            SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eff)); \
        KERNEL<<<grid, CO_TILE, lds_eff, st>>>(d_x.p, d_y.p, d_tl.p, d_tv.p, d_thr.p, d_cell.p, p);                             \
    } while (0)
            if (list) {
                if (fma) SQGR_CO((k_cooccur_fast<true, true>)); else SQGR_CO((k_cooccur_fast<false, true>));
            } else if (fast) {
                if (fma) SQGR_CO(k_cooccur_fast<true>); else SQGR_CO(k_cooccur_fast<false>);
            } else {
                if (fma) SQGR_CO(k_cooccur<true>); else SQGR_CO(k_cooccur<false>);
            }
#undef SQGR_CO
            SQGR_HIP(hipGetLastError());
        }
        std::vector<unsigned long long> tmp((size_t)K * K * L_eff);
        SQGR_HIP(hipMemcpyAsync(tmp.data(), d_out.p, tmp.size() * 8, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
        for (size_t ab = 0; ab < (size_t)K * K; ++ab) {
            unsigned long long run = 0;
            for (int g = 0; g < L_eff; ++g) {
                run += tmp[ab * L_eff + g];
                cum_host[ab * L + s0 + g] = run;
            }
        }
    }
    // ---- cumulative counts back in the caller's threshold order (NaN thresholds count nothing)
    for (size_t ab = 0; ab < (size_t)K * K; ++ab)
        for (int g = 0; g < L; ++g) out_counts[ab * L + order[g]] = (g < L_all) ? (int64_t)cum_host[ab * L + g] : 0;
    return SQGR_OK;
}
