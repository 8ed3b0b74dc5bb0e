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

#include <algorithm>

namespace sqgr {

constexpr int RP_TILE = 256;
constexpr int RP_CHUNK = 32;

template <int METRIC>
__device__ __forceinline__ double metric_dist(double xi, double yi, double xj, double yj) {
    const double dx = xi - xj, dy = yi - yj;
    if (METRIC == 0) return __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));  // reduced euclidean, every op rounded
    if (METRIC == 1) return fabs(dx) + fabs(dy);                               // manhattan
    return fmax(fabs(dx), fabs(dy));                                           // chebyshev
}

template <int METRIC>
__global__ __launch_bounds__(RP_TILE) void k_pair_hist(const double* __restrict__ xs, const double* __restrict__ ys, int64_t m,
                                                       const double* __restrict__ thr, int S, int T,
                                                       unsigned long long* __restrict__ out) {
    extern __shared__ unsigned char smem_raw[];
    double* s_thr = reinterpret_cast<double*>(smem_raw);                 // [S]
    uint32_t* hist = reinterpret_cast<uint32_t*>(s_thr + S);             // [S][256]
    const int t = threadIdx.x;
    const int ti = blockIdx.x;
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
    for (int64_t j = 0; j < nr; ++j) {
        double d = metric_dist<METRIC>(xi, yi, rx[j], ry[j]);  // rx[j]: wave-uniform scalar load
        if (d < best[KMAX - 1]) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {  // sorted insertion by compare-exchange down the register list
                const double lo = fmin(best[s], d), hi = fmax(best[s], d);
                best[s] = lo;
                d = hi;
            }
        }
    }
    if (active) {
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s < k) out[(size_t)q * k + s] = best[s];
    }
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
    x.resize((size_t)std::max<int64_t>(m, 1));
    y.resize((size_t)std::max<int64_t>(m, 1));
    for (int64_t i = 0; i < m; ++i) {
        x[i] = xy[2 * i];
        y[i] = xy[2 * i + 1];
    }
    return SQGR_OK;
}

}  // namespace sqgr

using namespace sqgr;

extern "C" {

int sqgr_pair_counts(sqgr_ctx* ctx, const double* xy, int64_t m, const double* thr, int32_t S, int32_t metric,
                     int64_t* out_counts) {
    SQGR_REQUIRE(ctx && thr && out_counts && (xy || m == 0), "null argument");
    SQGR_REQUIRE(m >= 0 && S >= 1 && metric >= 0 && metric <= 2, "bad argument m=%lld S=%d metric=%d", (long long)m, S, metric);
    for (int s = 1; s < S; ++s) SQGR_REQUIRE(thr[s - 1] <= thr[s], "thresholds must be ascending");
    const size_t lds = (size_t)S * 8 + (size_t)S * RP_TILE * 4;
    if (lds > 160 * 1024) {
        set_error("S=%d radii need %zu bytes of LDS (> 160 KiB)", S, lds);
        return SQGR_ERR_UNSUPPORTED;
    }
    for (int s = 0; s < S; ++s) out_counts[s] = 0;
    if (m < 2) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));
    std::vector<double> x, y;
    split_xy(xy, m, x, y);
    DevBuf<double> dx, dy, dthr;
    DevBuf<unsigned long long> dout;
    SQGR_TRY(dx.alloc((size_t)m));
    SQGR_TRY(dy.alloc((size_t)m));
    SQGR_TRY(dthr.alloc((size_t)S));
    SQGR_TRY(dout.alloc((size_t)S));
    hipStream_t st = ctx->stream;
    SQGR_HIP(hipMemcpyAsync(dx.p, x.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dy.p, y.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dthr.p, thr, (size_t)S * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemsetAsync(dout.p, 0, (size_t)S * 8, st));
    const int T = (int)ceil_div(m, RP_TILE);
    dim3 grid((unsigned)T, (unsigned)ceil_div(T, RP_CHUNK));
    {
        LaunchTimer t(ctx, "ripley_pair_hist");
#define SQGR_PH(M)                                                                                                         \
    do {                                                                                                                    \
        if (lds > 64 * 1024)                                                                                                \
            SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_hist<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_pair_hist<M><<<grid, RP_TILE, lds, st>>>(dx.p, dy.p, m, dthr.p, S, T, dout.p);                                   \
    } while (0)
        if (metric == 0) SQGR_PH(0); else if (metric == 1) SQGR_PH(1); else SQGR_PH(2);
#undef SQGR_PH
        SQGR_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> h((size_t)S);
    SQGR_HIP(hipMemcpyAsync(h.data(), dout.p, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    unsigned long long run = 0;
    for (int s = 0; s < S; ++s) {
        run += h[s];
        out_counts[s] = (int64_t)run;
    }
    return SQGR_OK;
}

int sqgr_knn_dist(sqgr_ctx* ctx, const double* query, int64_t nq, const double* ref, int64_t nr, int32_t k, int32_t metric,
                  double* out) {
    SQGR_REQUIRE(ctx && out && (query || nq == 0) && (ref || nr == 0), "null argument");
    SQGR_REQUIRE(nq >= 0 && nr >= 0 && metric >= 0 && metric <= 2, "bad argument");
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
    split_xy(ref, nr, rx, ry);
    DevBuf<double> dqx, dqy, drx, dry, dout;
    SQGR_TRY(dqx.alloc((size_t)nq));
    SQGR_TRY(dqy.alloc((size_t)nq));
    SQGR_TRY(drx.alloc((size_t)nr));
    SQGR_TRY(dry.alloc((size_t)nr));
    SQGR_TRY(dout.alloc((size_t)nq * k));
    hipStream_t st = ctx->stream;
    SQGR_HIP(hipMemcpyAsync(dqx.p, qx.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dqy.p, qy.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(drx.p, rx.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(dry.p, ry.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    {
        LaunchTimer t(ctx, "ripley_knn");
        if (metric == 0) SQGR_TRY(launch_knn<0>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
        else if (metric == 1) SQGR_TRY(launch_knn<1>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
        else SQGR_TRY(launch_knn<2>(ctx, dqx.p, dqy.p, nq, drx.p, dry.p, nr, k, dout.p));
    }
    SQGR_HIP(hipMemcpyAsync(out, dout.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

}  // extern "C"
