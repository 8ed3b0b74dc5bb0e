// libsqgr: spatial graph construction — exact k-nearest-neighbour and fixed-radius neighbour search in 2-D
// (the producer of the CSR graph the hot path consumes).
//
// Reference semantics (/root/reference/src/squidpy/gr/neighbors.py):
//   :196-199, :402-405  NearestNeighbors(n_neighbors=k, metric="euclidean").fit(coords).kneighbors()
//                       -> the k nearest OTHER samples of every sample (self excluded by index), ascending distance
//   :252-255            NearestNeighbors(radius=r).fit(coords).radius_neighbors()  -> all other samples with dist <= r
// sklearn answers these with a KD-tree on the host; distances are sqrt of the coordinate-wise accumulated squared
// distance (no FMA), and the radius test is done on squared distances (rdist <= r*r).
//
// MI355X design: a uniform grid ("cell list") over the bounding box with ~2 points per cell is built by a counting
// sort on the host (O(n)); one thread per query walks the cells ring by ring around its own cell, keeping the k
// best (d2, index) pairs in registers, and stops as soon as the k-th best is strictly closer than the unexplored
// region.  Exact (not approximate) and deterministic: ties are broken by the smaller sample index.
#include "sqgr_common.h"
#include "sqgr_grid.h"

#include <algorithm>
#include <cmath>

namespace sqgr {

__device__ __forceinline__ double sqdist(double ax, double ay, double bx, double by) {
    const double dx = ax - bx, dy = ay - by;
    return __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));  // -ffp-contract=off: never fused
}

// lexicographic (d2, index) order
__device__ __forceinline__ bool closer(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }

// sx/sy/sid: points sorted by cell; cell_start[c] .. cell_start[c+1]: members of cell c (row-major cy*gx + cx)
template <int KMAX>
__global__ __launch_bounds__(128) void k_knn_grid(CellGrid g, const double* __restrict__ sx, const double* __restrict__ sy,
                                                  const int32_t* __restrict__ sid, const int32_t* __restrict__ cell_start,
                                                  int64_t n, int k, int32_t* __restrict__ out_idx, double* __restrict__ out_d2) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // queries in cell order: neighbouring threads
    if (t >= n) return;                                                // walk neighbouring cells (cache locality)
    const double qx = sx[t], qy = sy[t];
    const int qid = sid[t];
    int cx, cy;
    cell_of(g, qx, qy, cx, cy);
    double bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        bd[s] = __builtin_inf();
        bi[s] = 0x7fffffff;
    }
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
                    const int id = sid[p];
                    if (id == qid) continue;
                    double d = sqdist(qx, qy, sx[p], sy[p]);
                    int di = id;
                    if (closer(d, di, bd[KMAX - 1], bi[KMAX - 1])) {
#pragma unroll
                        for (int s = 0; s < KMAX; ++s) {  // sorted insertion: compare-exchange down the register list
                            const bool sw = closer(d, di, bd[s], bi[s]);
                            const double td = sw ? bd[s] : d;
                            const int ti = sw ? bi[s] : di;
                            bd[s] = sw ? d : bd[s];
                            bi[s] = sw ? di : bi[s];
                            d = td;
                            di = ti;
                        }
                    }
                }
            }
        }
        // everything not yet visited lies outside the block of cells [cx-r, cx+r] x [cy-r, cy+r]
        const double xl = g.x0 + (double)(cx - r) * g.h, xh = g.x0 + (double)(cx + r + 1) * g.h;
        const double yl = g.y0 + (double)(cy - r) * g.h, yh = g.y0 + (double)(cy + r + 1) * g.h;
        const double m = fmin(fmin(qx - xl, xh - qx), fmin(qy - yl, yh - qy)) - 1e-9 * g.h;  // slack for cell rounding
        double kth = __builtin_inf();  // k-th best so far (select chain: a dynamic register index would spill)
#pragma unroll
        for (int s = 0; s < KMAX; ++s) kth = (s == k - 1) ? bd[s] : kth;
        if (m > 0.0 && kth < m * m) break;  // strict: an unexplored point at exactly the k-th distance could win a tie
    }
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
        if (s < k) {
            out_idx[(size_t)qid * k + s] = bi[s];
            out_d2[(size_t)qid * k + s] = bd[s];
        }
}

// fixed radius: COUNT == true  -> counts[qid] = #neighbours;  COUNT == false -> fill rows at offsets indptr[qid]
template <bool COUNT>
__global__ __launch_bounds__(128) void k_radius_grid(CellGrid g, const double* __restrict__ sx, const double* __restrict__ sy,
                                                     const int32_t* __restrict__ sid, const int32_t* __restrict__ cell_start,
                                                     int64_t n, double r2, int reach, int64_t* __restrict__ counts,
                                                     const int64_t* __restrict__ indptr, int32_t* __restrict__ out_idx,
                                                     double* __restrict__ out_d2) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double qx = sx[t], qy = sy[t];
    const int qid = sid[t];
    int cx, cy;
    cell_of(g, qx, qy, cx, cy);
    int64_t cnt = 0;
    const int64_t base = COUNT ? 0 : indptr[qid];
    for (int yy = max(cy - reach, 0); yy <= min(cy + reach, g.gy - 1); ++yy)
        for (int xx = max(cx - reach, 0); xx <= min(cx + reach, g.gx - 1); ++xx) {
            const int c = yy * g.gx + xx;
            for (int p = cell_start[c]; p < cell_start[c + 1]; ++p) {
                const int id = sid[p];
                if (id == qid) continue;
                const double d = sqdist(qx, qy, sx[p], sy[p]);
                if (d <= r2) {
                    if (!COUNT) {
                        out_idx[base + cnt] = id;
                        out_d2[base + cnt] = d;
                    }
                    ++cnt;
                }
            }
        }
    if (COUNT) counts[qid] = cnt;
}

// counting sort of the points into ~n/target cells
int build_grid(const double* xy, int64_t n, double target_per_cell, double min_h, HostGrid& out) {
    double x0 = xy[0], x1 = xy[0], y0 = xy[1], y1 = xy[1];
    for (int64_t i = 0; i < n; ++i) {
        const double x = xy[2 * i], y = xy[2 * i + 1];
        if (!std::isfinite(x) || !std::isfinite(y)) {
            set_error("coordinate %lld is not finite", (long long)i);
            return SQGR_ERR_INVALID;
        }
        x0 = std::min(x0, x); x1 = std::max(x1, x);
        y0 = std::min(y0, y); y1 = std::max(y1, y);
    }
    const double w = std::max(x1 - x0, 1e-300), hgt = std::max(y1 - y0, 1e-300);
    double h = std::sqrt(w * hgt * target_per_cell / (double)n);
    h = std::max(h, std::max(w, hgt) / 4096.0);  // at most 4096 x 4096 cells
    h = std::max(h, min_h);
    if (!(h > 0.0) || !std::isfinite(h)) h = 1.0;
    CellGrid g;
    g.x0 = x0; g.y0 = y0; g.h = h; g.inv_h = 1.0 / h;
    g.gx = std::max(1, (int)std::floor(w / h) + 1);
    g.gy = std::max(1, (int)std::floor(hgt / h) + 1);
    out.g = g;
    const size_t ncell = (size_t)g.gx * g.gy;
    out.cell_start.assign(ncell + 1, 0);
    std::vector<int32_t> cell((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        int cx = std::min(std::max((int)std::floor((xy[2 * i] - x0) * g.inv_h), 0), g.gx - 1);
        int cy = std::min(std::max((int)std::floor((xy[2 * i + 1] - y0) * g.inv_h), 0), g.gy - 1);
        cell[i] = cy * g.gx + cx;
        out.cell_start[cell[i] + 1]++;
    }
    for (size_t c = 0; c < ncell; ++c) out.cell_start[c + 1] += out.cell_start[c];
    out.sx.resize((size_t)n); out.sy.resize((size_t)n); out.sid.resize((size_t)n);
    std::vector<int32_t> fill(out.cell_start.begin(), out.cell_start.end() - 1);
    for (int64_t i = 0; i < n; ++i) {  // stable: members of a cell stay in index order
        const int32_t p = fill[cell[i]]++;
        out.sx[p] = xy[2 * i];
        out.sy[p] = xy[2 * i + 1];
        out.sid[p] = (int32_t)i;
    }
    return SQGR_OK;
}

}  // namespace sqgr

using namespace sqgr;

extern "C" {

int sqgr_knn_self(sqgr_ctx* ctx, const double* xy, int64_t n, int32_t k, int32_t* out_idx, double* out_d2) {
    SQGR_REQUIRE(ctx && xy && out_idx && out_d2, "null argument");
    SQGR_REQUIRE(n >= 1 && n < (int64_t)0x7fffffff, "n=%lld out of range", (long long)n);
    SQGR_REQUIRE(k >= 1 && k < n, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld, n_samples = %lld",
                 k + 1, (long long)n, (long long)n);  // sklearn queries k+1 and drops the sample itself
    if (k > 64) {
        set_error("n_neighbors=%d > 64 is not supported by the register-resident kNN search", k);
        return SQGR_ERR_UNSUPPORTED;
    }
    SQGR_HIP(hipSetDevice(ctx->device));
    HostGrid hg;
    SQGR_TRY(build_grid(xy, n, 2.0, 0.0, hg));
    DevGrid dg;
    hipStream_t st = ctx->stream;
    SQGR_TRY(dg.upload(hg, st));
    DevBuf<int32_t> d_idx;
    DevBuf<double> d_d2;
    SQGR_TRY(d_idx.alloc((size_t)n * k));
    SQGR_TRY(d_d2.alloc((size_t)n * k));
    {
        LaunchTimer t(ctx, "neighbors_knn_grid");
        const unsigned grid = (unsigned)ceil_div(n, 128);
#define SQGR_KNN(KM) k_knn_grid<KM><<<grid, 128, 0, st>>>(hg.g, dg.sx.p, dg.sy.p, dg.sid.p, dg.cell_start.p, n, k, d_idx.p, d_d2.p)
        if (k <= 4) SQGR_KNN(4); else if (k <= 8) SQGR_KNN(8); else if (k <= 16) SQGR_KNN(16); else if (k <= 32) SQGR_KNN(32); else SQGR_KNN(64);
#undef SQGR_KNN
        SQGR_HIP(hipGetLastError());
    }
    SQGR_HIP(hipMemcpyAsync(out_idx, d_idx.p, (size_t)n * k * 4, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipMemcpyAsync(out_d2, d_d2.p, (size_t)n * k * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_radius_self(sqgr_ctx* ctx, const double* xy, int64_t n, double radius, int64_t* out_indptr, int32_t* out_idx,
                     double* out_d2, int64_t capacity) {
    SQGR_REQUIRE(ctx && xy && out_indptr, "null argument");
    SQGR_REQUIRE(n >= 1 && n < (int64_t)0x7fffffff, "n=%lld out of range", (long long)n);
    SQGR_REQUIRE(radius >= 0.0 && std::isfinite(radius), "radius must be finite and >= 0");
    SQGR_HIP(hipSetDevice(ctx->device));
    HostGrid hg;
    SQGR_TRY(build_grid(xy, n, 2.0, radius / 8.0, hg));  // never more than ~17 x 17 cells per query
    const int reach = (int)std::ceil(radius * hg.g.inv_h) + 1;
    DevGrid dg;
    hipStream_t st = ctx->stream;
    SQGR_TRY(dg.upload(hg, st));
    const double r2 = radius * radius;  // sklearn: EuclideanDistance._dist_to_rdist
    DevBuf<int64_t> d_cnt, d_ptr;
    SQGR_TRY(d_cnt.alloc((size_t)n));
    const unsigned grid = (unsigned)ceil_div(n, 128);
    {
        LaunchTimer t(ctx, "neighbors_radius_count");
        k_radius_grid<true><<<grid, 128, 0, st>>>(hg.g, dg.sx.p, dg.sy.p, dg.sid.p, dg.cell_start.p, n, r2, reach, d_cnt.p, nullptr,
                                                 nullptr, nullptr);
        SQGR_HIP(hipGetLastError());
    }
    std::vector<int64_t> cnt((size_t)n);
    SQGR_HIP(hipMemcpyAsync(cnt.data(), d_cnt.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    out_indptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) out_indptr[i + 1] = out_indptr[i] + cnt[i];
    if (!out_idx || !out_d2) return SQGR_OK;  // counting pass only
    const int64_t nnz = out_indptr[n];
    SQGR_REQUIRE(capacity >= nnz, "capacity %lld < %lld neighbours", (long long)capacity, (long long)nnz);
    if (nnz == 0) return SQGR_OK;
    DevBuf<int32_t> d_idx;
    DevBuf<double> d_d2;
    SQGR_TRY(d_ptr.alloc((size_t)n + 1));
    SQGR_TRY(d_idx.alloc((size_t)nnz));
    SQGR_TRY(d_d2.alloc((size_t)nnz));
    SQGR_HIP(hipMemcpyAsync(d_ptr.p, out_indptr, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    {
        LaunchTimer t(ctx, "neighbors_radius_fill");
        k_radius_grid<false><<<grid, 128, 0, st>>>(hg.g, dg.sx.p, dg.sy.p, dg.sid.p, dg.cell_start.p, n, r2, reach, nullptr, d_ptr.p,
                                                  d_idx.p, d_d2.p);
        SQGR_HIP(hipGetLastError());
    }
    SQGR_HIP(hipMemcpyAsync(out_idx, d_idx.p, (size_t)nnz * 4, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipMemcpyAsync(out_d2, d_d2.p, (size_t)nnz * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

}  // extern "C"
