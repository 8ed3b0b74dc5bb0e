// libsqgr: Ripley's K/L pair counting and F/G nearest-neighbour distances in float64.
//
// Reference semantics (/root/reference/src/squidpy/gr/_ripley.py):
//   :212-227  _l_function: KDTree(points).two_point_correlation(points, support, dualtree=True) - m
//             = #{ordered pairs i != j : dist_ij <= r} for every r in support (cumulative), dist as sklearn's KDTree
//             computes it: rdist accumulated coordinate by coordinate (no FMA), then sqrt.
//   :144-150, :163-169  NearestNeighbors(n_neighbors=k).kneighbors(queries): the k smallest distances per query.
//
// MI355X design: brute force beats tree traversal here — the point sets are per-cluster (tens of thousands of
// points) and the arithmetic is 5 float64 ops per pair.  Pair counting reuses the co-occurrence structure (256-point
// tiles, tj points through scalar loads, private LDS histogram columns, each unordered tile pair once); the kNN
// sweep keeps the k best squared distances of one query per thread in registers.
// `sqrt(d2) <= r` is decided WITHOUT a square root: the caller passes, per radius, the largest float64 t with
// fl(sqrt(t)) <= r, so `d2 <= t` is the same predicate bit for bit.
#include "sqgr_common.h"
#include "sqgr_grid.h"

#include <algorithm>
#include <cmath>

namespace sqgr {

constexpr int RP_TILE = 256;
constexpr int RP_CHUNK = 8;  // tj tiles per block (round 3: 32 -> 8: four times the blocks per launch — a cluster of config 4 is 131 tiles)

template <int METRIC>
__device__ __forceinline__ double metric_dist(double xi, double yi, double xj, double yj) {
    const double dx = xi - xj, dy = yi - yj;
    if (METRIC == 0) return __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));  // reduced euclidean, every op rounded
    if (METRIC == 1) return fabs(dx) + fabs(dy);                               // manhattan
    if (METRIC == 2) return fmax(fabs(dx), fabs(dy));                          // chebyshev
    // canberra — the parameter-free BallTree metric a metric STRING can also name (the reference hands `metric` straight to
    // NearestNeighbors, gr/_ripley.py:144,148): sum |x - y| / (|x| + |y|), 0/0 terms skipped, sklearn's per-coordinate loop.
    // (braycurtis was tried too: it is not a metric, sklearn's ball tree prunes with the triangle inequality anyway and returns
    //  neighbours an exact search does not — nothing to be bit-compatible with.)
    const double ex = fabs(xi) + fabs(xj), ey = fabs(yi) + fabs(yj);
    double d = 0.0;
    if (ex > 0.0) d += fabs(dx) / ex;
    if (ey > 0.0) d += fabs(dy) / ey;
    return d;
}

// Point sets of one launch (blockIdx.z): set z has set_m[z] points stored from tile set_tile0[z] on in the packed, tile-padded
// coordinate arrays; its counts go to out + z * S.  (Ripley's L evaluates one set per cluster and per simulation: one launch for
// all of them instead of 30 + 100 small ones.)
struct PairSets {
    const int64_t* m;
    const int32_t* tile0;
};

template <int METRIC>
__global__ __launch_bounds__(RP_TILE) void k_pair_hist(const double* __restrict__ xs_all, const double* __restrict__ ys_all, PairSets sets,
                                                       const double* __restrict__ thr, int S,
                                                       unsigned long long* __restrict__ out_all) {
    extern __shared__ unsigned char smem_raw[];
    double* s_thr = reinterpret_cast<double*>(smem_raw);                 // [S]
    uint32_t* hist = reinterpret_cast<uint32_t*>(s_thr + S);             // [S][256]
    const int t = threadIdx.x;
    const int64_t m = sets.m[blockIdx.z];
    const int T = (int)((m + RP_TILE - 1) / RP_TILE);
    const double* xs = xs_all + (size_t)sets.tile0[blockIdx.z] * RP_TILE;
    const double* ys = ys_all + (size_t)sets.tile0[blockIdx.z] * RP_TILE;
    unsigned long long* out = out_all + (size_t)blockIdx.z * S;
    const int ti = blockIdx.x;
    if (ti >= T) return;
    const int tj0 = max(ti, (int)blockIdx.y * RP_CHUNK);
    const int tj1 = min(T, ((int)blockIdx.y + 1) * RP_CHUNK);
    if (tj0 >= tj1) return;
    for (int i = t; i < S; i += RP_TILE) s_thr[i] = thr[i];
    for (int i = t; i < S * RP_TILE; i += RP_TILE) hist[i] = 0;
    __syncthreads();
    const int64_t gi = (int64_t)ti * RP_TILE + t;
    const bool active = gi < m;
    const double xi = active ? xs[gi] : 0.0, yi = active ? ys[gi] : 0.0;
    uint32_t* my = hist + t;
    for (int tj = tj0; tj < tj1; ++tj) {
        const int64_t j0 = (int64_t)tj * RP_TILE;
        const int vj = (int)min<int64_t>(RP_TILE, m - j0);
        const double* __restrict__ xj = xs + j0;  // wave-uniform: scalar loads
        const double* __restrict__ yj = ys + j0;
        const bool diag = (tj == ti);
        // ordered pairs: a diagonal tile contributes each ordered pair once, an off-diagonal tile pair twice
        const uint32_t w = diag ? 1u : 2u;
        if (active) {
            for (int j = 0; j < vj; ++j) {
                const double d = metric_dist<METRIC>(xi, yi, xj[j], yj[j]);
                int lo = 0, hi = S;  // first threshold index with d <= thr  (S = none); NaN -> S
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (d <= s_thr[mid]) hi = mid; else lo = mid + 1;
                }
                if (lo < S && !(diag && j == t)) atomicAdd(my + lo * RP_TILE, w);
            }
        }
    }
    __syncthreads();
    for (int g = t; g < S; g += RP_TILE) {
        unsigned long long s = 0;
        for (int k = 0; k < RP_TILE; ++k) s += hist[g * RP_TILE + ((k + t) & (RP_TILE - 1))];
        if (s) atomicAdd(&out[g], s);
    }
}

// Branch-free variant (same scheme as k_cooccur_fast): a lookup table over the metric value gives a lower bound g of the
// bin with true bin <= g + 2; two compares against the adjacent thresholds finish it.  Full batches of off-diagonal
// tiles with finite coordinates skip all validity tests (out-of-range pairs land in RP_TRASH write-only rows).
// Histogram columns are per LANE (64), shared by the block's four waves through the LDS atomics they are anyway: 13 KB
// instead of 54 KB at 50 radii, so five blocks instead of two share a CU (round 3: the kernel was latency-bound at
// 8 waves per CU — a cluster of config 4 is only 650 blocks).  A column receives at most 4 waves x RP_CHUNK tiles x 256 x 2 counts.
constexpr int RP_BATCH = 8;
constexpr int RP_TRASH = 3;
constexpr int RP_CELLS_MIN = 1024;
constexpr int RP_CELLS_MAX = 32768;

template <int METRIC>
__global__ __launch_bounds__(RP_TILE) void k_pair_hist_fast(const double* __restrict__ xs_all, const double* __restrict__ ys_all,
                                                            PairSets sets, const double* __restrict__ thr, int S,
                                                            const uint16_t* __restrict__ cell, int ncells, double inv_cell,
                                                            int finite, unsigned long long* __restrict__ out_all) {
    extern __shared__ unsigned char smem_raw[];
    // [S + 2][32]: every threshold (and two +inf sentinels) 32 times, copy c in bank pair c — lane l reads copy l & 31, so the
    // threshold reads of a wave (random bins) never meet in a bank (tools/ubench_ds_mix.hip: ds_read2_b64 of the plain array ~11 clk)
    double* s_thr = reinterpret_cast<double*>(smem_raw);
    constexpr int HC = 64;                                                     // histogram columns
    uint32_t* hist = reinterpret_cast<uint32_t*>(s_thr + (S + 2) * 32);        // [S + RP_TRASH][HC]
    uint16_t* s_cell = reinterpret_cast<uint16_t*>(hist + (S + RP_TRASH) * HC);  // [ncells]
    const int t = threadIdx.x;
    const int64_t m = sets.m[blockIdx.z];
    const int T = (int)((m + RP_TILE - 1) / RP_TILE);
    const double* xs = xs_all + (size_t)sets.tile0[blockIdx.z] * RP_TILE;
    const double* ys = ys_all + (size_t)sets.tile0[blockIdx.z] * RP_TILE;
    unsigned long long* out = out_all + (size_t)blockIdx.z * S;
    const int ti = blockIdx.x;
    if (ti >= T) return;
    const int tj0 = max(ti, (int)blockIdx.y * RP_CHUNK);
    const int tj1 = min(T, ((int)blockIdx.y + 1) * RP_CHUNK);
    if (tj0 >= tj1) return;
    for (int i = t; i < (S + 2) * 32; i += RP_TILE) s_thr[i] = (i >> 5) < S ? thr[i >> 5] : __builtin_inf();
    for (int i = t; i < (S + RP_TRASH) * HC; i += RP_TILE) hist[i] = 0;
    for (int i = t; i < ncells; i += RP_TILE) s_cell[i] = cell[i];
    __syncthreads();
    const int64_t gi = (int64_t)ti * RP_TILE + t;
    const bool active = gi < m;
    const double xi = active ? xs[gi] : 0.0, yi = active ? ys[gi] : 0.0;
    const int cmax = ncells - 1;
    uint32_t* my = hist + (t & (HC - 1));
    const int l32 = t & 31;
    for (int tj = tj0; tj < tj1; ++tj) {
        const int64_t j0g = (int64_t)tj * RP_TILE;
        const int vj = (int)min<int64_t>(RP_TILE, m - j0g);
        const double* __restrict__ xj = xs + j0g;  // wave-uniform: scalar loads (arrays are padded to whole tiles)
        const double* __restrict__ yj = ys + j0g;
        const bool diag = (tj == ti);
        const uint32_t w = diag ? 1u : 2u;  // a diagonal tile yields each ordered pair once, an off-diagonal tile pair twice
        auto batch = [&](int j0, auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
            double d[RP_BATCH];
            int g[RP_BATCH];
#pragma unroll
            for (int u = 0; u < RP_BATCH; ++u) {
                d[u] = metric_dist<METRIC>(xi, yi, xj[j0 + u], yj[j0 + u]);
                int cellv = (int)(d[u] * inv_cell);  // saturating conversion; NaN -> 0
                if constexpr (CHECKED) cellv = max(cellv, 0);
                g[u] = s_cell[min(cellv, cmax)];
            }
            double t0[RP_BATCH], t1[RP_BATCH];  // fetched before the first LDS atomic (reads cannot move across it)
#pragma unroll
            for (int u = 0; u < RP_BATCH; ++u) {
                t0[u] = s_thr[g[u] * 32 + l32];
                t1[u] = s_thr[g[u] * 32 + 32 + l32];
            }
#pragma unroll
            for (int u = 0; u < RP_BATCH; ++u) {
                const int c0 = !(d[u] <= t0[u]), c1 = !(d[u] <= t1[u]);
                const int gg = g[u] + c0 + c1;
                if constexpr (CHECKED) {
                    const bool ok = (gg < S) & (j0 + u < vj) & !(diag && j0 + u == t) & (d[u] == d[u]);
                    atomicAdd(my + min(gg, S - 1) * HC, ok ? w : 0u);
                } else {
                    atomicAdd(my + gg * HC, w);
                }
            }
        };
        if (active) {
            if (diag || !finite) {
                for (int j0 = 0; j0 < vj; j0 += RP_BATCH) batch(j0, std::true_type{});
            } else {
                const int jfull = vj & ~(RP_BATCH - 1);
                for (int j0 = 0; j0 < jfull; j0 += RP_BATCH) batch(j0, std::false_type{});
                if (jfull < vj) batch(jfull, std::true_type{});
            }
        }
    }
    __syncthreads();
    for (int g = t; g < S; g += RP_TILE) {
        unsigned long long s = 0;
        for (int k = 0; k < HC; ++k) s += hist[g * HC + ((k + t) & (HC - 1))];
        if (s) atomicAdd(&out[g], s);
    }
}

// k smallest metric distances (euclidean: squared) of every query to the reference set, ascending.
template <int METRIC, int KMAX>
__global__ __launch_bounds__(256) void k_knn(const double* __restrict__ qx, const double* __restrict__ qy, int64_t nq,
                                             const double* __restrict__ rx, const double* __restrict__ ry, int64_t nr, int k,
                                             double* __restrict__ out) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool active = q < nq;
    const double xi = active ? qx[q] : 0.0, yi = active ? qy[q] : 0.0;
    double best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = __builtin_inf();
    auto offer = [&](double d) {
        if (d < best[KMAX - 1]) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {  // sorted insertion by compare-exchange down the register list
                const double lo = fmin(best[s], d), hi = fmax(best[s], d);
                best[s] = lo;
                d = hi;
            }
        }
    };
    // references arrive through wave-uniform scalar loads; batches of 8 so that one s_load_dwordx16 pair feeds eight
    // distance evaluations instead of one load (and one SMEM latency) per reference
    constexpr int KB = 8;
    const int64_t nr_full = nr & ~(int64_t)(KB - 1);
    for (int64_t j = 0; j < nr_full; j += KB) {
        double d[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) d[u] = metric_dist<METRIC>(xi, yi, rx[j + u], ry[j + u]);
#pragma unroll
        for (int u = 0; u < KB; ++u) offer(d[u]);
    }
    for (int64_t j = nr_full; j < nr; ++j) offer(metric_dist<METRIC>(xi, yi, rx[j], ry[j]));
    if (active) {
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s < k) out[(size_t)q * k + s] = best[s];
    }
}

// The same sweep for query points that stay on the device, followed by numpy's histogram of the k distances
// (`np.histogram(d, bins=edges)`: bin i counts edges[i] <= d < edges[i+1], the last bin also d == edges[S-1]; values
// outside [edges[0], edges[S-1]] and NaN are dropped).  Queries whose label equals `exclude` do not take part (Ripley's G:
// every point NOT in the cluster against the cluster's points, gr/_ripley.py:163-169).
template <int METRIC, int KMAX>
__global__ __launch_bounds__(256) void k_knn_hist(const double* __restrict__ qx, const double* __restrict__ qy,
                                                  const int32_t* __restrict__ qlabel, int exclude, int64_t nq,
                                                  const double* __restrict__ rx, const double* __restrict__ ry, int64_t nr, int k,
                                                  const double* __restrict__ edges, int S, unsigned long long* __restrict__ out) {
    extern __shared__ unsigned char knn_smem[];
    double* s_edges = reinterpret_cast<double*>(knn_smem);             // [S]
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_edges + S);       // [S - 1]
    for (int i = threadIdx.x; i < S; i += 256) s_edges[i] = edges[i];
    for (int i = threadIdx.x; i < S - 1; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool active = q < nq && (exclude < 0 || qlabel[q] != exclude);
    const double xi = active ? qx[q] : 0.0, yi = active ? qy[q] : 0.0;
    double best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = __builtin_inf();
    auto offer = [&](double d) {
        if (d < best[KMAX - 1]) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const double lo = fmin(best[s], d), hi = fmax(best[s], d);
                best[s] = lo;
                d = hi;
            }
        }
    };
    constexpr int KB = 8;
    const int64_t nr_full = nr & ~(int64_t)(KB - 1);
    for (int64_t j = 0; j < nr_full; j += KB) {
        double d[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) d[u] = metric_dist<METRIC>(xi, yi, rx[j + u], ry[j + u]);
#pragma unroll
        for (int u = 0; u < KB; ++u) offer(d[u]);
    }
    for (int64_t j = nr_full; j < nr; ++j) offer(metric_dist<METRIC>(xi, yi, rx[j], ry[j]));
    if (active) {
        const double e_first = s_edges[0], e_last = s_edges[S - 1];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if (s < k) {
                const double v = (METRIC == 0) ? __dsqrt_rn(best[s]) : best[s];  // euclidean keeps squared distances
                if (v >= e_first && v <= e_last) {   // false for NaN
                    int lo = 0, hi = S - 1;          // largest i in [0, S-2] with edges[i] <= v (v == e_last -> S-2)
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_edges[mid] <= v) lo = mid; else hi = mid;
                    }
                    atomicAdd(&s_hist[lo], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S - 1; i += 256)
        if (s_hist[i]) atomicAdd(&out[i], (unsigned long long)s_hist[i]);
}

// The same two sweeps through a CELL LIST of the reference points (round 3; the brute-force kernels above cost
// n_queries x n_refs distance evaluations — 1e12 for Ripley's G at 1e6 points).  One thread per query walks the cells of
// the reference grid ring by ring around the query's own cell, keeps the k best metric values in registers and stops as
// soon as the k-th best is strictly closer than anything outside the block of cells visited so far; for all three
// metrics a point outside the block [xl, xh] x [yl, yh] is at least m = (distance from the query to the block's boundary)
// away (L2 and L1 >= Linf >= m).  Queries may lie outside the grid (F draws them in the convex hull of ALL points): their
// cell is the clamped one, m is negative until the block reaches them, and the walk ends at the latest when every cell has
// been seen.  The k smallest values are the same multiset whatever the visiting order, so the result equals the brute-force
// sweep bit for bit.  HIST: numpy's histogram of the distances instead of the distances (see k_knn_hist).
template <int METRIC, int KMAX, bool HIST>
__global__ __launch_bounds__(128) void k_knn_cells(CellGrid g, const double* __restrict__ sx, const double* __restrict__ sy,
                                                   const int32_t* __restrict__ cell_start, const double* __restrict__ qx,
                                                   const double* __restrict__ qy, const int32_t* __restrict__ qlabel, int exclude,
                                                   int64_t nq, int k, double* __restrict__ out, const double* __restrict__ edges, int S,
                                                   unsigned long long* __restrict__ hist_out) {
    extern __shared__ unsigned char knn_smem[];
    double* s_edges = reinterpret_cast<double*>(knn_smem);        // [S]      (HIST only)
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_edges + S);  // [S - 1]
    if constexpr (HIST) {
        for (int i = threadIdx.x; i < S; i += blockDim.x) s_edges[i] = edges[i];
        for (int i = threadIdx.x; i < S - 1; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
    }
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool active = q < nq && (!HIST || exclude < 0 || qlabel[q] != exclude);
    double best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = __builtin_inf();
    if (active) {
        const double xi = qx[q], yi = qy[q];
        int cx, cy;
        cell_of(g, xi, yi, cx, cy);
        const int rmax = max(max(cx, g.gx - 1 - cx), max(cy, g.gy - 1 - cy));
        for (int r = 0; r <= rmax; ++r) {
            const int ylo = cy - r, yhi = cy + r;
            for (int yy = max(ylo, 0); yy <= min(yhi, g.gy - 1); ++yy) {
                const bool edge_row = (yy == ylo) || (yy == yhi);
                const int step = edge_row ? 1 : 2 * r;  // interior rows of the ring: only the two end cells
                for (int xx = cx - r; xx <= cx + r; xx += (step > 0 ? step : 1)) {
                    if (xx < 0 || xx >= g.gx) continue;
                    const int c = yy * g.gx + xx;
                    for (int p = cell_start[c]; p < cell_start[c + 1]; ++p) {
                        double d = metric_dist<METRIC>(xi, yi, sx[p], sy[p]);
                        if (d < best[KMAX - 1]) {
#pragma unroll
                            for (int s = 0; s < KMAX; ++s) {
                                const double lo = fmin(best[s], d), hi = fmax(best[s], d);
                                best[s] = lo;
                                d = hi;
                            }
                        }
                    }
                }
            }
            const double xl = g.x0 + (double)(cx - r) * g.h, xh = g.x0 + (double)(cx + r + 1) * g.h;
            const double yl = g.y0 + (double)(cy - r) * g.h, yh = g.y0 + (double)(cy + r + 1) * g.h;
            const double m = fmin(fmin(xi - xl, xh - xi), fmin(yi - yl, yh - yi)) - 1e-9 * g.h;  // slack for cell rounding
            double kth = __builtin_inf();
#pragma unroll
            for (int s = 0; s < KMAX; ++s) kth = (s == k - 1) ? best[s] : kth;
            if (m > 0.0 && kth < (METRIC == 0 ? m * m : m)) break;  // strict: ties at the k-th distance are harmless either way
        }
        if constexpr (!HIST) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
                if (s < k) out[(size_t)q * k + s] = best[s];
        } else {
            const double e_first = s_edges[0], e_last = s_edges[S - 1];
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                if (s < k) {
                    const double v = (METRIC == 0) ? __dsqrt_rn(best[s]) : best[s];
                    if (v >= e_first && v <= e_last) {
                        int lo = 0, hi = S - 1;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_edges[mid] <= v) lo = mid; else hi = mid;
                        }
                        atomicAdd(&s_hist[lo], 1u);
                    }
                }
            }
        }
    }
    if constexpr (HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < S - 1; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&hist_out[i], (unsigned long long)s_hist[i]);
    }
}

// reference sets from this size on go through the cell list (below it the brute-force sweep is as fast and needs no grid)
constexpr int64_t KNN_GRID_MIN_REFS = 512;

template <int METRIC, bool HIST>
static int launch_knn_cells(sqgr_ctx* ctx, const HostGrid& hg, const DevGrid& dg, const double* qx, const double* qy, const int32_t* qlabel,
                            int exclude, int64_t nq, int k, double* out, const double* edges, int S, unsigned long long* hist_out) {
    const unsigned grid = (unsigned)ceil_div(nq, 128);
    const size_t lds = HIST ? (size_t)S * 8 + (size_t)S * 4 : 0;
    hipStream_t st = ctx->stream;
#define SQGR_KC(KM) \
    k_knn_cells<METRIC, KM, HIST><<<grid, 128, lds, st>>>(hg.g, dg.sx.p, dg.sy.p, dg.cell_start.p, qx, qy, qlabel, exclude, nq, k, out, edges, S, hist_out)
    if (k <= 1) SQGR_KC(1);
    else if (k <= 2) SQGR_KC(2);
    else if (k <= 4) SQGR_KC(4);
    else if (k <= 8) SQGR_KC(8);
    else SQGR_KC(16);
#undef SQGR_KC
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

static bool all_finite(const double* xy, int64_t n) {
    for (int64_t i = 0; i < 2 * n; ++i)
        if (!std::isfinite(xy[i])) return false;
    return true;
}

template <int METRIC>
static int launch_knn(sqgr_ctx* ctx, const double* qx, const double* qy, int64_t nq, const double* rx, const double* ry, int64_t nr,
                      int k, double* out) {
    const unsigned grid = (unsigned)ceil_div(nq, 256);
    hipStream_t st = ctx->stream;
    if (k <= 1) k_knn<METRIC, 1><<<grid, 256, 0, st>>>(qx, qy, nq, rx, ry, nr, k, out);
    else if (k <= 2) k_knn<METRIC, 2><<<grid, 256, 0, st>>>(qx, qy, nq, rx, ry, nr, k, out);
    else if (k <= 4) k_knn<METRIC, 4><<<grid, 256, 0, st>>>(qx, qy, nq, rx, ry, nr, k, out);
    else if (k <= 8) k_knn<METRIC, 8><<<grid, 256, 0, st>>>(qx, qy, nq, rx, ry, nr, k, out);
    else k_knn<METRIC, 16><<<grid, 256, 0, st>>>(qx, qy, nq, rx, ry, nr, k, out);
    SQGR_HIP(hipGetLastError());
    return SQGR_OK;
}

static int split_xy(const double* xy, int64_t m, std::vector<double>& x, std::vector<double>& y) {
    const size_t padded = (size_t)ceil_div(std::max<int64_t>(m, 1), RP_TILE) * RP_TILE;  // whole tiles, zero padded
    x.assign(padded, 0.0);
    y.assign(padded, 0.0);
    for (int64_t i = 0; i < m; ++i) {
        x[i] = xy[2 * i];
        y[i] = xy[2 * i + 1];
    }
    return SQGR_OK;
}

}  // namespace sqgr

using namespace sqgr;

struct sqgr_points {  // a point set resident on the device (coordinates split into x / y, optional integer label per point)
    sqgr_ctx* ctx = nullptr;
    int64_t n = 0;
    bool has_label = false;
    DevBuf<double> x, y;
    DevBuf<int32_t> label;
};

extern "C" {

int sqgr_pair_counts_batch(sqgr_ctx* ctx, const double* xy, const int64_t* offsets, int32_t n_sets, const double* thr, int32_t S,
                           int32_t metric, int64_t* out_counts) {
    SQGR_REQUIRE(ctx && thr && out_counts && offsets && n_sets >= 0, "null argument or n_sets < 0");
    SQGR_REQUIRE(S >= 1 && metric >= 0 && metric <= 2, "bad argument S=%d metric=%d", S, metric);
    for (int s = 1; s < S; ++s) SQGR_REQUIRE(thr[s - 1] <= thr[s], "thresholds must be ascending");
    SQGR_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
    for (int z = 0; z < n_sets; ++z) SQGR_REQUIRE(offsets[z] <= offsets[z + 1], "offsets are not ascending at set %d", z);
    SQGR_REQUIRE(n_sets <= 65535, "at most 65535 point sets per call, found %d", n_sets);
    const int64_t total = n_sets ? offsets[n_sets] : 0;
    SQGR_REQUIRE(xy || total == 0, "xy is NULL");
    const size_t lds = (size_t)S * 8 + (size_t)S * RP_TILE * 4;
    if (lds > 160 * 1024) {
        set_error("S=%d radii need %zu bytes of LDS (> 160 KiB)", S, lds);
        return SQGR_ERR_UNSUPPORTED;
    }
    for (int64_t i = 0; i < (int64_t)n_sets * S; ++i) out_counts[i] = 0;
    if (n_sets == 0 || total == 0) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));
    // pack: every set padded with zeros to whole tiles
    std::vector<int64_t> set_m((size_t)n_sets);
    std::vector<int32_t> tile0((size_t)n_sets + 1, 0);
    int64_t max_m = 0;
    for (int z = 0; z < n_sets; ++z) {
        set_m[z] = offsets[z + 1] - offsets[z];
        tile0[z + 1] = tile0[z] + (int32_t)ceil_div(std::max<int64_t>(set_m[z], 1), RP_TILE);
        max_m = std::max(max_m, set_m[z]);
    }
    if (max_m < 2) return SQGR_OK;
    const size_t mp = (size_t)tile0[n_sets] * RP_TILE;
    std::vector<double> x(mp, 0.0), y(mp, 0.0);
    bool finite = true;
    for (int z = 0; z < n_sets; ++z) {
        const size_t base = (size_t)tile0[z] * RP_TILE;
        for (int64_t i = 0; i < set_m[z]; ++i) {
            const double px = xy[2 * (offsets[z] + i)], py = xy[2 * (offsets[z] + i) + 1];
            x[base + i] = px;
            y[base + i] = py;
            finite = finite && std::isfinite(px) && std::isfinite(py);
        }
    }
    // lookup table over the metric value for the branch-free kernel: cell c -> lower bound of the bin of every value in it
    double tmax = 0.0;
    for (int s2 = 0; s2 < S; ++s2)
        if (std::isfinite(thr[s2])) tmax = std::max(tmax, thr[s2]);
    std::vector<uint16_t> cell;
    int ncells = RP_CELLS_MIN;
    double inv_cell = 0.0;
    bool fast = false;
    const size_t lds_fast_fixed = (size_t)(S + 2) * 32 * 8 + (size_t)(S + RP_TRASH) * 64 * 4;
    if (tmax > 0.0 && S < 65000) {
        for (;; ncells *= 2) {
            inv_cell = (double)ncells / tmax;
            if (!std::isfinite(inv_cell) || lds_fast_fixed + (size_t)ncells * 2 > 160 * 1024) break;
            cell.assign((size_t)ncells, 0);
            int worst = 0;
            for (int c = 0; c < ncells; ++c) {
                // every value landing in cell c lies in [(c-1)/inv_cell, (c+2)/inv_cell): a full cell of slack on both sides
                // absorbs the rounding of value * inv_cell
                const double lo = (c >= 1) ? ((double)(c - 1) / inv_cell) * (1.0 - 1e-9) : -1.0;
                const double hi = (c == ncells - 1) ? (double)INFINITY : ((double)(c + 2) / inv_cell) * (1.0 + 1e-9);
                int g = 0;
                while (g < S && thr[g] < lo) ++g;
                int gh = g;
                while (gh < S && thr[gh] < hi) ++gh;
                cell[c] = (uint16_t)g;
                worst = std::max(worst, gh - g);
            }
            if (worst <= 2) {
                fast = true;
                break;
            }
            if (ncells * 2 > RP_CELLS_MAX) break;
        }
    }
    struct { double* p; } dx, dy, dthr;  // context scratch: this entry point may run once per cluster and per simulation
    struct { unsigned long long* p; } dout;
    struct { uint16_t* p; } dcell;
    struct { int64_t* p; } dm;
    struct { int32_t* p; } dt0;
    SQGR_TRY(ctx->scratch_get(0, mp * 8, reinterpret_cast<void**>(&dx.p)));
    SQGR_TRY(ctx->scratch_get(1, mp * 8, reinterpret_cast<void**>(&dy.p)));
    SQGR_TRY(ctx->scratch_get(4, (size_t)RP_CELLS_MAX * 2, reinterpret_cast<void**>(&dcell.p)));
    SQGR_TRY(ctx->scratch_get(2, (size_t)S * 8, reinterpret_cast<void**>(&dthr.p)));
    SQGR_TRY(ctx->scratch_get(3, (size_t)n_sets * S * 8, reinterpret_cast<void**>(&dout.p)));
    SQGR_TRY(ctx->scratch_get(5, (size_t)n_sets * 8, reinterpret_cast<void**>(&dm.p)));
    SQGR_TRY(ctx->scratch_get(6, ((size_t)n_sets + 1) * 4, reinterpret_cast<void**>(&dt0.p)));
    hipStream_t st = ctx->stream;
    SQGR_HIP(hipMemcpyAsync(dx.p, x.data(), mp * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dy.p, y.data(), mp * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dthr.p, thr, (size_t)S * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dm.p, set_m.data(), (size_t)n_sets * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dt0.p, tile0.data(), ((size_t)n_sets + 1) * 4, hipMemcpyHostToDevice, st));
    if (fast) SQGR_HIP(hipMemcpyAsync(dcell.p, cell.data(), (size_t)ncells * 2, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemsetAsync(dout.p, 0, (size_t)n_sets * S * 8, st));
    const int T = (int)ceil_div(max_m, RP_TILE);
    const dim3 grid((unsigned)T, (unsigned)ceil_div(T, RP_CHUNK), (unsigned)n_sets);
    const PairSets sets{dm.p, dt0.p};
    if (fast) {
        LaunchTimer t(ctx, "ripley_pair_hist_fast");
        const size_t lds_fast = lds_fast_fixed + (size_t)ncells * 2;
#define SQGR_PHF(M)                                                                                                        \
    do {                                                                                                                    \
        if (lds_fast > 64 * 1024)                                                                                           \
            SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_hist_fast<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast)); \
        k_pair_hist_fast<M><<<grid, RP_TILE, lds_fast, st>>>(dx.p, dy.p, sets, dthr.p, S, dcell.p, ncells, inv_cell, finite ? 1 : 0, dout.p); \
    } while (0)
        if (metric == 0) SQGR_PHF(0); else if (metric == 1) SQGR_PHF(1); else SQGR_PHF(2);
#undef SQGR_PHF
        SQGR_HIP(hipGetLastError());
    } else {
        LaunchTimer t(ctx, "ripley_pair_hist");
#define SQGR_PH(M)                                                                                                         \
    do {                                                                                                                    \
        if (lds > 64 * 1024)                                                                                                \
            SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_hist<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_pair_hist<M><<<grid, RP_TILE, lds, st>>>(dx.p, dy.p, sets, dthr.p, S, dout.p);                                   \
    } while (0)
        if (metric == 0) SQGR_PH(0); else if (metric == 1) SQGR_PH(1); else SQGR_PH(2);
#undef SQGR_PH
        SQGR_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> h((size_t)n_sets * S);
    SQGR_HIP(hipMemcpyAsync(h.data(), dout.p, (size_t)n_sets * S * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    for (int z = 0; z < n_sets; ++z) {
        unsigned long long run = 0;
        for (int s = 0; s < S; ++s) {
            run += h[(size_t)z * S + s];
            out_counts[(size_t)z * S + s] = (int64_t)run;
        }
    }
    return SQGR_OK;
}

int sqgr_pair_counts(sqgr_ctx* ctx, const double* xy, int64_t m, const double* thr, int32_t S, int32_t metric,
                     int64_t* out_counts) {
    SQGR_REQUIRE(m >= 0, "bad argument m=%lld", (long long)m);
    SQGR_REQUIRE(xy || m == 0, "null argument");
    const int64_t offsets[2] = {0, m};
    return sqgr_pair_counts_batch(ctx, xy, offsets, 1, thr, S, metric, out_counts);
}

int sqgr_knn_dist(sqgr_ctx* ctx, const double* query, int64_t nq, const double* ref, int64_t nr, int32_t k, int32_t metric,
                  double* out) {
    SQGR_REQUIRE(ctx && out && (query || nq == 0) && (ref || nr == 0), "null argument");
    SQGR_REQUIRE(nq >= 0 && nr >= 0 && metric >= 0 && metric <= 3, "bad argument");
    SQGR_REQUIRE(k >= 1 && k <= nr, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld, n_samples = %lld",
                 k, (long long)nr, (long long)nq);
    if (k > 16) {
        set_error("n_neighbors=%d > 16 is not supported by the register-resident kNN sweep", k);
        return SQGR_ERR_UNSUPPORTED;
    }
    if (nq == 0) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));
    std::vector<double> qx, qy, rx, ry;
    split_xy(query, nq, qx, qy);
    if (metric <= 2 && nr >= KNN_GRID_MIN_REFS && all_finite(ref, nr)) {  // cell list over the reference points (the Lp metrics)
        HostGrid hg;
        SQGR_TRY(build_grid(ref, nr, 2.0, 0.0, hg));
        DevGrid dg;
        hipStream_t st = ctx->stream;
        SQGR_TRY(dg.upload(hg, st));
        struct { double* p; } dqx, dqy, dout;
        SQGR_TRY(ctx->scratch_get(0, (size_t)nq * 8, reinterpret_cast<void**>(&dqx.p)));
        SQGR_TRY(ctx->scratch_get(1, (size_t)nq * 8, reinterpret_cast<void**>(&dqy.p)));
        SQGR_TRY(ctx->scratch_get(4, (size_t)nq * k * 8, reinterpret_cast<void**>(&dout.p)));
        SQGR_HIP(hipMemcpyAsync(dqx.p, qx.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(dqy.p, qy.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
        {
            LaunchTimer t(ctx, "ripley_knn_cells");
            if (metric == 0) SQGR_TRY((launch_knn_cells<0, false>(ctx, hg, dg, dqx.p, dqy.p, nullptr, -1, nq, k, dout.p, nullptr, 0, nullptr)));
            else if (metric == 1) SQGR_TRY((launch_knn_cells<1, false>(ctx, hg, dg, dqx.p, dqy.p, nullptr, -1, nq, k, dout.p, nullptr, 0, nullptr)));
            else SQGR_TRY((launch_knn_cells<2, false>(ctx, hg, dg, dqx.p, dqy.p, nullptr, -1, nq, k, dout.p, nullptr, 0, nullptr)));
        }
        SQGR_HIP(hipMemcpyAsync(out, dout.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));  // also keeps `dg` alive until the kernel is done
        return SQGR_OK;
    }
    split_xy(ref, nr, rx, ry);
    struct { double* p; } dqx, dqy, drx, dry, dout;  // context scratch (see sqgr_pair_counts)
    SQGR_TRY(ctx->scratch_get(0, (size_t)nq * 8, reinterpret_cast<void**>(&dqx.p)));
    SQGR_TRY(ctx->scratch_get(1, (size_t)nq * 8, reinterpret_cast<void**>(&dqy.p)));
    SQGR_TRY(ctx->scratch_get(2, (size_t)nr * 8, reinterpret_cast<void**>(&drx.p)));
    SQGR_TRY(ctx->scratch_get(3, (size_t)nr * 8, reinterpret_cast<void**>(&dry.p)));
    SQGR_TRY(ctx->scratch_get(4, (size_t)nq * k * 8, reinterpret_cast<void**>(&dout.p)));
    hipStream_t st = ctx->stream;
    SQGR_HIP(hipMemcpyAsync(dqx.p, qx.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dqy.p, qy.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(drx.p, rx.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dry.p, ry.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    {
        LaunchTimer t(ctx, "ripley_knn");
        if (metric == 0) SQGR_TRY(launch_knn<0>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
        else if (metric == 1) SQGR_TRY(launch_knn<1>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
        else if (metric == 2) SQGR_TRY(launch_knn<2>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
        else SQGR_TRY(launch_knn<3>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
    }
    SQGR_HIP(hipMemcpyAsync(out, dout.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_points_create(sqgr_ctx* ctx, const double* xy, const int32_t* labels, int64_t n, sqgr_points** out) {
    SQGR_REQUIRE(ctx && xy && out && n > 0, "null argument or n <= 0");
    *out = nullptr;
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_points* p = new sqgr_points();
    p->ctx = ctx;
    p->n = n;
    p->has_label = labels != nullptr;
    std::vector<double> x, y;
    split_xy(xy, n, x, y);
    int rc = SQGR_OK;
    if ((rc = p->x.alloc((size_t)n)) || (rc = p->y.alloc((size_t)n)) || (rc = p->label.alloc((size_t)n))) {
        delete p;
        return rc;
    }
    hipError_t e = hipMemcpy(p->x.p, x.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->y.p, y.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && labels) e = hipMemcpy(p->label.p, labels, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("point upload failed: %s", hipGetErrorString(e));
        delete p;
        return SQGR_ERR_HIP;
    }
    *out = p;
    return SQGR_OK;
}

int sqgr_points_destroy(sqgr_points* p) {
    if (!p) return SQGR_OK;
    (void)hipSetDevice(p->ctx->device);
    delete p;
    return SQGR_OK;
}

int sqgr_knn_hist(sqgr_ctx* ctx, const sqgr_points* queries, int32_t exclude_label, const double* ref, int64_t nr, int32_t k,
                  int32_t metric, const double* edges, int32_t S, int64_t* out_counts) {
    SQGR_REQUIRE(ctx && queries && ref && edges && out_counts, "null argument");
    SQGR_REQUIRE(queries->ctx == ctx, "points belong to a different context");
    SQGR_REQUIRE(S >= 2 && S <= 8192 && metric >= 0 && metric <= 3, "bad argument S=%d metric=%d", S, metric);
    SQGR_REQUIRE(exclude_label < 0 || queries->has_label, "points were created without labels");
    SQGR_REQUIRE(k >= 1 && k <= nr, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld, n_samples = %lld", k,
                 (long long)nr, (long long)queries->n);
    for (int s2 = 1; s2 < S; ++s2) SQGR_REQUIRE(edges[s2 - 1] <= edges[s2], "bin edges must be ascending");
    if (k > 16) {
        set_error("n_neighbors=%d > 16 is not supported by the register-resident kNN sweep", k);
        return SQGR_ERR_UNSUPPORTED;
    }
    SQGR_HIP(hipSetDevice(ctx->device));
    if (metric <= 2 && nr >= KNN_GRID_MIN_REFS && all_finite(ref, nr)) {  // cell list over the reference points (the Lp metrics)
        HostGrid hg;
        SQGR_TRY(build_grid(ref, nr, 2.0, 0.0, hg));
        DevGrid dg;
        hipStream_t st = ctx->stream;
        SQGR_TRY(dg.upload(hg, st));
        struct { double* p; } dedges;
        struct { unsigned long long* p; } dout;
        SQGR_TRY(ctx->scratch_get(2, (size_t)S * 8, reinterpret_cast<void**>(&dedges.p)));
        SQGR_TRY(ctx->scratch_get(3, (size_t)S * 8, reinterpret_cast<void**>(&dout.p)));
        SQGR_HIP(hipMemcpyAsync(dedges.p, edges, (size_t)S * 8, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemsetAsync(dout.p, 0, (size_t)S * 8, st));
        {
            LaunchTimer t(ctx, "ripley_knn_hist_cells");
            const sqgr_points* qs = queries;
            if (metric == 0) SQGR_TRY((launch_knn_cells<0, true>(ctx, hg, dg, qs->x.p, qs->y.p, qs->label.p, exclude_label, qs->n, k, nullptr, dedges.p, S, dout.p)));
            else if (metric == 1) SQGR_TRY((launch_knn_cells<1, true>(ctx, hg, dg, qs->x.p, qs->y.p, qs->label.p, exclude_label, qs->n, k, nullptr, dedges.p, S, dout.p)));
            else SQGR_TRY((launch_knn_cells<2, true>(ctx, hg, dg, qs->x.p, qs->y.p, qs->label.p, exclude_label, qs->n, k, nullptr, dedges.p, S, dout.p)));
        }
        std::vector<unsigned long long> hc((size_t)S);
        SQGR_HIP(hipMemcpyAsync(hc.data(), dout.p, (size_t)(S - 1) * 8, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
        for (int s2 = 0; s2 < S - 1; ++s2) out_counts[s2] = (int64_t)hc[s2];
        return SQGR_OK;
    }
    std::vector<double> rx, ry;
    split_xy(ref, nr, rx, ry);
    struct { double* p; } drx, dry, dedges;
    struct { unsigned long long* p; } dout;
    SQGR_TRY(ctx->scratch_get(0, (size_t)nr * 8, reinterpret_cast<void**>(&drx.p)));
    SQGR_TRY(ctx->scratch_get(1, (size_t)nr * 8, reinterpret_cast<void**>(&dry.p)));
    SQGR_TRY(ctx->scratch_get(2, (size_t)S * 8, reinterpret_cast<void**>(&dedges.p)));
    SQGR_TRY(ctx->scratch_get(3, (size_t)S * 8, reinterpret_cast<void**>(&dout.p)));
    hipStream_t st = ctx->stream;
    SQGR_HIP(hipMemcpyAsync(drx.p, rx.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dry.p, ry.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dedges.p, edges, (size_t)S * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemsetAsync(dout.p, 0, (size_t)S * 8, st));
    {
        LaunchTimer t(ctx, "ripley_knn_hist");
        const unsigned grid = (unsigned)ceil_div(queries->n, 256);
        const size_t lds = (size_t)S * 8 + (size_t)S * 4;
#define SQGR_KH(M, KM) k_knn_hist<M, KM><<<grid, 256, lds, st>>>(queries->x.p, queries->y.p, queries->label.p, exclude_label, queries->n, drx.p, dry.p, nr, k, dedges.p, S, dout.p)
#define SQGR_KHM(M)                                \
    do {                                           \
        if (k <= 1) SQGR_KH(M, 1);                 \
        else if (k <= 2) SQGR_KH(M, 2);            \
        else if (k <= 4) SQGR_KH(M, 4);            \
        else if (k <= 8) SQGR_KH(M, 8);            \
        else SQGR_KH(M, 16);                       \
    } while (0)
        if (metric == 0) SQGR_KHM(0); else if (metric == 1) SQGR_KHM(1); else if (metric == 2) SQGR_KHM(2); else SQGR_KHM(3);
#undef SQGR_KHM
#undef SQGR_KH
        SQGR_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> h((size_t)S);
    SQGR_HIP(hipMemcpyAsync(h.data(), dout.p, (size_t)(S - 1) * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    for (int s2 = 0; s2 < S - 1; ++s2) out_counts[s2] = (int64_t)h[s2];
    return SQGR_OK;
}

}  // extern "C"
