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
};

template <bool FMA>
__device__ __forceinline__ float dist2(float xi, float yi, float xj, float yj) {
    const float dx = xi - xj, dy = yi - yj;
    if (FMA) return fmaf(dx, dx, dy * dy);
    return __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));  // numpy / un-contracted semantics: every op rounded
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
template <bool FMA>
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

    const int ti = blockIdx.x * p.shard_count + p.shard_index;
    if (ti >= p.T) return;
    const int tj0 = max(ti, (int)blockIdx.y * CO_CHUNK_TILES);
    const int tj1 = min(p.T, ((int)blockIdx.y + 1) * CO_CHUNK_TILES);
    if (tj0 >= tj1) return;

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

    int cur_b = -1;
    for (int tj = tj0; tj < tj1; ++tj) {
        const int b = tile_label[tj];
        if (b != cur_b) {
            if (cur_b >= 0) co_flush<CO_HC>(hist, L, p.K, a, cur_b, true, p.out);
            cur_b = b;
        }
        const int vj = tile_valid[tj];
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
        if (active) {
            if (tj == ti || !p.finite) {
                for (int j0 = 0; j0 < vj; j0 += CO_BATCH) batch(j0, std::true_type{});
            } else {
                const int jfull = vj & ~(CO_BATCH - 1);
                for (int j0 = 0; j0 < jfull; j0 += CO_BATCH) batch(j0, std::false_type{});
                if (jfull < vj) batch(jfull, std::true_type{});
            }
        }
        if (tj == ti) co_flush<CO_HC>(hist, L, p.K, a, a, false, p.out);  // diagonal tile: ordered pairs complete, credit (a,a) once
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
    std::vector<float> xs((size_t)T * CO_TILE, 0.f), ys((size_t)T * CO_TILE, 0.f);
    std::vector<int32_t> tile_label((size_t)T), tile_valid((size_t)T);
    std::vector<int64_t> fill((size_t)K, 0);
    bool all_finite = true;
    for (int64_t i = 0; i < n; ++i) {
        const int k = labels[i];
        const int64_t pos = tile0[k] * CO_TILE + fill[k]++;
        xs[pos] = x[i];
        ys[pos] = y[i];
        all_finite = all_finite && std::isfinite(x[i]) && std::isfinite(y[i]);
    }
    for (int k = 0; k < K; ++k)
        for (int64_t tt = tile0[k]; tt < tile0[k + 1]; ++tt) {
            tile_label[tt] = k;
            const int64_t left = cnt[k] - (tt - tile0[k]) * CO_TILE;
            tile_valid[tt] = (int32_t)std::min<int64_t>(left, CO_TILE);
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
        CoParams p{inv_cell, ncells, (int)T, L_eff, K, shard_index, shard_count, all_finite ? 1 : 0, d_out.p};
        const size_t lds_eff = fast ? (size_t)(L_eff + CO_TRASH) * CO_HC * 4 + (size_t)(L_eff + 2) * 32 * 4 + (size_t)ncells * 2 : lds_fixed + (size_t)ncells * 2;
        dim3 grid((unsigned)ceil_div(T, shard_count), (unsigned)ceil_div(T, CO_CHUNK_TILES));
        {
            LaunchTimer tm(ctx, fast ? (fma ? "cooccur_pairs_fast_fma" : "cooccur_pairs_fast") : (fma ? "cooccur_pairs_fma" : "cooccur_pairs"));
#define SQGR_CO(KERNEL)                                                                                                          \
    do {                                                                                                                         \
        if (lds_eff > 64 * 1024)                                                                                                 \
            SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eff)); \
        KERNEL<<<grid, CO_TILE, lds_eff, st>>>(d_x.p, d_y.p, d_tl.p, d_tv.p, d_thr.p, d_cell.p, p);                             \
    } while (0)
            if (fast) {
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
