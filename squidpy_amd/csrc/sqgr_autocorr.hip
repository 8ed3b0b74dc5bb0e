// libsqgr: Moran's I / Geary's C with row-permutation tests on the device-resident CSR graph.
//
// Reference semantics (/root/reference/src/squidpy/gr/_ppatterns.py):
//   :216       score = func(g, vals)                    func = scanpy.metrics.morans_i | gearys_c  (third party)
//   :258-280   per permutation p: idx = rng.permutation(N); score_perms[p] = func(g[idx, :], vals)
// i.e. the ROWS of the weight matrix are permuted, the values stay in place.  With z = x - mean(x):
//   Moran   I  = N/W * sum_i z_i (G z)_i / sum z^2            I_p = N/W * sum_i z_i y[idx_i] / sum z^2,  y = G z
//   Geary   C  = (N-1) sum_ij w_ij (z_i - z_j)^2 / (2 W sum z^2)
//           C_p = (N-1) [ sum_i z_i^2 r[idx_i] - 2 sum_i z_i y[idx_i] + sum_k q_k ] / (2 W sum z^2),
//                 r = G 1 (row sums), q = G z^2
// so 1000 permuted SpMVs per gene collapse into ONE SpMV plus 1000 gather-dots.
//
// Layout: genes are cut into tiles of 64 (one wavefront wide); Zt/Yt/Qt are stored [tile][spot][64] float64, so a
// "row" is 512 contiguous bytes and every gather y[idx_i] is one fully coalesced wave load.  The permutation kernel
// is launched tile-major so that the ~N*512 B working set of the tiles in flight stays in the 256 MB Infinity Cache.
// All reductions are two-stage with a fixed order => bit-reproducible run to run.
#include "sqgr_common.h"
#include "sqgr_rng.h"
#include "sqgr_pcg.h"

#include <cstdlib>

namespace sqgr {

constexpr int GT = 64;         // genes per tile
constexpr int PERM_TILE = 32;  // permutations per block (8 per wave)
constexpr uint32_t AUTOCORR_STREAM = 0x5A17u;  // "library" word of the Philox counter: separate stream from nhood

// ---- per-gene statistics of a staged gene-major block X[gc][n]: mean and is-constant flag
__global__ __launch_bounds__(256) void k_gene_stats(const double* __restrict__ X, int64_t n, double* __restrict__ mean,
                                                    uint8_t* __restrict__ isconst, int64_t g0) {
    __shared__ double ssum[256];
    __shared__ int sdiff[256];
    const int g = blockIdx.x;
    const double* x = X + (size_t)g * n;
    const double first = x[0];
    double s = 0.0;
    int diff = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double v = x[i];
        s += v;
        diff |= (v != first);
    }
    ssum[threadIdx.x] = s;
    sdiff[threadIdx.x] = diff;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            ssum[threadIdx.x] += ssum[threadIdx.x + o];
            sdiff[threadIdx.x] |= sdiff[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mean[g0 + g] = ssum[0] / (double)n;
        isconst[g0 + g] = sdiff[0] ? 0 : 1;
    }
}

// ---- cell-major input D[i][G_all] (what AnnData's X is) -> the staged gene-major block X[gl][i] of genes g0..g0+gc
__global__ __launch_bounds__(256) void k_cells_to_genes(const double* __restrict__ D, int64_t ld, int64_t n, int64_t g0, int gc,
                                                        double* __restrict__ X) {
    __shared__ double tile[GT][GT + 1];
    const int64_t i0 = (int64_t)blockIdx.x * GT;
    const int gb = blockIdx.y * GT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < GT; r += 4) {  // r: spot inside tile, tx: gene
        const int64_t i = i0 + r;
        tile[r][tx] = (i < n && gb + tx < gc) ? D[(size_t)i * ld + g0 + gb + tx] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < GT; r += 4) {  // r: gene inside tile, tx: spot
        const int64_t i = i0 + tx;
        if (gb + r < gc && i < n) X[(size_t)(gb + r) * n + i] = tile[tx][r];
    }
}

// ---- Zt[tile][i][gl] = X[g][i] - mean[g]   (64 x 64 tile transpose through LDS; gc is a multiple of 64 or the tail)
__global__ __launch_bounds__(256) void k_center_transpose(const double* __restrict__ X, int64_t n, int gc, int64_t g0,
                                                          const double* __restrict__ mean, double* __restrict__ Zt) {
    __shared__ double tile[GT][GT + 1];
    const int64_t i0 = (int64_t)blockIdx.x * GT;
    const int gb = blockIdx.y * GT;  // gene offset inside the staged block
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < GT; r += 4) {  // r: gene inside tile, tx: spot
        const int g = gb + r;
        const int64_t i = i0 + tx;
        tile[r][tx] = (g < gc && i < n) ? X[(size_t)g * n + i] - mean[g0 + g] : 0.0;
    }
    __syncthreads();
    const int64_t tile_id = (g0 + gb) / GT;
    for (int r = ty; r < GT; r += 4) {  // r: spot inside tile, tx: gene
        const int64_t i = i0 + r;
        if (i < n) Zt[((size_t)tile_id * n + i) * GT + tx] = tile[tx][r];
    }
}

// ---- Yt = G Zt, Qt = G Zt^2 (row i, 64 genes per wave; CSR metadata is wave-uniform => scalar loads)
__global__ __launch_bounds__(256) void k_spmv_tiles(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                    const double* __restrict__ data, int64_t n, const double* __restrict__ Zt,
                                                    double* __restrict__ Yt, double* __restrict__ Qt) {
    const int gl = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* Z = Zt + (size_t)blockIdx.y * n * GT;
    double y = 0.0, q = 0.0;
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
        const double w = data[e];
        const double zc = Z[(size_t)indices[e] * GT + gl];
        y += w * zc;
        q += w * (zc * zc);
    }
    const size_t o = ((size_t)blockIdx.y * n + i) * GT + gl;
    Yt[o] = y;
    Qt[o] = q;
}

__global__ void k_rowsum(const int64_t* __restrict__ indptr, const double* __restrict__ data, int64_t n, double* __restrict__ rs) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) s += data[e];
    rs[i] = s;
}

// ---- column reductions over spots, stage 1: partial[tile][chunk][gl]
// MODE 0: sum A*B   1: sum A*A   2: sum A   3: Geary direct  sum_e w_e (z_i - z_col)^2
template <int MODE>
__global__ __launch_bounds__(256) void k_colsum(const double* __restrict__ A, const double* __restrict__ Bm, int64_t n, int R,
                                                const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                const double* __restrict__ data, double* __restrict__ partial) {
    __shared__ double red[4][GT];
    const int gl = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int chunk = blockIdx.x, tile = blockIdx.y;
    const int64_t per = (n + R - 1) / R;
    const int64_t i0 = chunk * per, i1 = min(n, i0 + per);
    const double* a = A + (size_t)tile * n * GT;
    const double* b = Bm ? Bm + (size_t)tile * n * GT : nullptr;
    double s = 0.0;
    for (int64_t i = i0 + sub; i < i1; i += 4) {
        const double av = a[(size_t)i * GT + gl];
        if (MODE == 0) s += av * b[(size_t)i * GT + gl];
        if (MODE == 1) s += av * av;
        if (MODE == 2) s += av;
        if (MODE == 3) {
            double t = 0.0;
            for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
                const double d = av - a[(size_t)indices[e] * GT + gl];
                t += data[e] * (d * d);
            }
            s += t;
        }
    }
    red[sub][gl] = s;
    __syncthreads();
    if (sub == 0) partial[((size_t)tile * R + chunk) * GT + gl] = (red[0][gl] + red[1][gl]) + (red[2][gl] + red[3][gl]);
}

__global__ void k_colsum_final(const double* __restrict__ partial, int R, int64_t G, double* __restrict__ out) {
    int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int64_t tile = g / GT;
    const int gl = (int)(g % GT);
    double s = 0.0;
    for (int c = 0; c < R; ++c) s += partial[((size_t)tile * R + c) * GT + gl];
    out[g] = s;
}

// ---- permutation indices from the device generator: idx[p][i] = pi_p(i)
__global__ __launch_bounds__(256) void k_perm_indices(uint64_t seed, int64_t perm0, int64_t n, FeistelDomain dom,
                                                      int32_t* __restrict__ idx) {
    __shared__ uint32_t rk[8];
    const int64_t p = blockIdx.y;
    if (threadIdx.x == 0) {
        uint32_t k[8];
        round_keys(seed, (uint64_t)(perm0 + p), AUTOCORR_STREAM, k);
        for (int j = 0; j < 8; ++j) rk[j] = k[j];
    }
    __syncthreads();
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    idx[(size_t)p * n + i] = (int32_t)feistel_perm((uint32_t)i, dom, rk);
}

// ---- the hot kernel: partial sums of  S1[p][g] = sum_i z[i][g] * y[idx_p(i)][g]   (and, for Geary,
//      S2[p][g] = sum_i z[i][g]^2 * r[idx_p(i)])  over one row chunk.
// grid = (perm tiles, row chunks, gene tiles) — gene tile is the slowest grid dimension.
// block = 4 waves; wave w owns PERM_TILE/4 permutations of the block's PERM_TILE; lane = gene.
template <bool GEARY>
__global__ __launch_bounds__(256) void k_perm_dot(const double* __restrict__ Zt, const double* __restrict__ Yt,
                                                  const double* __restrict__ rowsum, const int32_t* __restrict__ idx, int64_t n,
                                                  int64_t nperm, int R, double* __restrict__ part1, double* __restrict__ part2) {
    constexpr int PW = PERM_TILE / 4;  // permutations per wave: each z row read serves PW gathers
    const int gl = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * PERM_TILE + wv * PW;
    if (p0 >= nperm) return;
    const int chunk = blockIdx.y, tile = blockIdx.z;
    const int64_t per = (n + R - 1) / R;
    const int64_t i0 = chunk * per, i1 = min(n, i0 + per);
    const double* Z = Zt + (size_t)tile * n * GT + gl;
    const double* Y = Yt + (size_t)tile * n * GT + gl;
    // clamp: the surplus permutations of the last wave recompute permutation nperm-1 and are not stored
    const int32_t* ix[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) ix[k] = idx + (size_t)min(p0 + k, nperm - 1) * n;
    double a[PW], b[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) a[k] = b[k] = 0.0;
#pragma unroll 2
    for (int64_t i = i0; i < i1; ++i) {
        const double z = Z[(size_t)i * GT];
        const double zz = z * z;
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int32_t row = ix[k][i];  // wave-uniform: scalar load
            a[k] = fma(z, Y[(size_t)row * GT], a[k]);
            if (GEARY) b[k] = fma(zz, rowsum[row], b[k]);  // (pre-gathering r[idx] into its own array measured slower)
        }
    }
    // partial layout [tile][p][chunk][gl]
    const size_t stride = (size_t)R * GT;
    double* o1 = part1 + ((size_t)tile * nperm * R + chunk) * GT + gl;
    double* o2 = GEARY ? part2 + ((size_t)tile * nperm * R + chunk) * GT + gl : nullptr;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        if (p0 + k < nperm) {
            o1[(size_t)(p0 + k) * stride] = a[k];
            if (GEARY) o2[(size_t)(p0 + k) * stride] = b[k];
        }
    }
}

// ---- stage 2 + statistic:  sims[p][g]
template <bool GEARY>
__global__ __launch_bounds__(256) void k_perm_final(const double* __restrict__ part1, const double* __restrict__ part2, int R,
                                                    int64_t nperm, int64_t G, int64_t n, double W, const double* __restrict__ z2ss,
                                                    const double* __restrict__ qsum, const uint8_t* __restrict__ isconst,
                                                    double* __restrict__ sims) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    if (g >= G) return;
    const int64_t tile = g / GT;
    const int gl = (int)(g % GT);
    const size_t base = (((size_t)tile * nperm + p) * R) * GT + gl;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < R; ++c) {
        s1 += part1[base + (size_t)c * GT];
        if (GEARY) s2 += part2[base + (size_t)c * GT];
    }
    double v;
    if (GEARY)
        v = ((double)(n - 1) * ((s2 - 2.0 * s1) + qsum[g])) / (2.0 * W * z2ss[g]);
    else
        v = (double)n / W * s1 / z2ss[g];
    sims[(size_t)p * G + g] = isconst[g] ? __builtin_nan("") : v;
}

__global__ void k_scores(int mode, int64_t G, int64_t n, double W, const double* __restrict__ num, const double* __restrict__ z2ss,
                         const uint8_t* __restrict__ isconst, double* __restrict__ out) {
    int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= G) return;
    double v = (mode == 0) ? (double)n / W * num[g] / z2ss[g] : ((double)(n - 1) * num[g]) / (2.0 * W * z2ss[g]);
    out[g] = isconst[g] ? __builtin_nan("") : v;
}

}  // namespace sqgr

using namespace sqgr;

struct sqgr_matrix {  // a dense row-major float64 matrix resident on the device (the expression matrix, uploaded once)
    sqgr_ctx* ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0;
    DevBuf<double> data;
};

struct sqgr_autocorr {
    sqgr_ctx* ctx = nullptr;
    const sqgr_graph* g = nullptr;
    int64_t n = 0, G = 0, ntiles = 0;
    double W = 0.0;
    int Rcol = 32;
    DevBuf<double> Zt, Yt, Qt, rowsum, mean, z2ss, qsum, tmpG, colpart;
    DevBuf<uint8_t> isconst;
    // permutation workspace
    DevBuf<int32_t> idx;
    DevBuf<double> part1, part2, sims;
    PcgWorkspace pcg_ws;          // numpy-stream permutations generated in place (sqgr_autocorr_perms_pcg64)
    DevBuf<uint64_t> pcg_states;
};

static int column_sum(sqgr_autocorr* h, int mode, const double* A, const double* B, double* out_dev) {
    sqgr_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    const int R = h->Rcol;
    dim3 grid(R, (unsigned)h->ntiles);
    const sqgr_graph* g = h->g;
    {
        LaunchTimer t(ctx, "autocorr_colsum");
        switch (mode) {
            case 0: k_colsum<0><<<grid, 256, 0, st>>>(A, B, h->n, R, nullptr, nullptr, nullptr, h->colpart.p); break;
            case 1: k_colsum<1><<<grid, 256, 0, st>>>(A, nullptr, h->n, R, nullptr, nullptr, nullptr, h->colpart.p); break;
            case 2: k_colsum<2><<<grid, 256, 0, st>>>(A, nullptr, h->n, R, nullptr, nullptr, nullptr, h->colpart.p); break;
            default: k_colsum<3><<<grid, 256, 0, st>>>(A, nullptr, h->n, R, g->indptr.p, g->indices.p, g->data.p, h->colpart.p); break;
        }
        SQGR_HIP(hipGetLastError());
        k_colsum_final<<<(unsigned)ceil_div(h->G, 256), 256, 0, st>>>(h->colpart.p, R, h->G, out_dev);
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
}

// vals: host block (gene-major, or cell-major when `cell_major`), or NULL when the features are columns
// [dev_col0, dev_col0 + G) of a matrix already resident on the device (dev_x[i * dev_ld + col])
static int autocorr_create(sqgr_ctx* ctx, const sqgr_graph* g, const double* vals, int64_t G, bool cell_major, sqgr_autocorr** out,
                           const double* dev_x = nullptr, int64_t dev_ld = 0, int64_t dev_col0 = 0) {
    SQGR_REQUIRE(ctx && g && (vals || dev_x) && out, "null argument");
    *out = nullptr;
    SQGR_REQUIRE(g->ctx == ctx, "graph belongs to a different context");
    SQGR_REQUIRE(g->has_data || g->nnz == 0, "graph was uploaded without edge weights");
    SQGR_REQUIRE(G >= 1, "G=%lld", (long long)G);
    SQGR_HIP(hipSetDevice(ctx->device));
    const int64_t n = g->n;
    sqgr_autocorr* h = new sqgr_autocorr();
    h->ctx = ctx;
    h->g = g;
    h->n = n;
    h->G = G;
    h->ntiles = ceil_div(G, GT);
    hipStream_t st = ctx->stream;
    int rc = SQGR_OK;
    auto fail = [&](int code) {
        delete h;
        return code;
    };
    const size_t tsz = (size_t)h->ntiles * n * GT;
    if ((rc = h->Zt.alloc(tsz)) || (rc = h->Yt.alloc(tsz)) || (rc = h->Qt.alloc(tsz)) || (rc = h->rowsum.alloc((size_t)n)) ||
        (rc = h->mean.alloc((size_t)G)) || (rc = h->z2ss.alloc((size_t)G)) || (rc = h->qsum.alloc((size_t)G)) ||
        (rc = h->tmpG.alloc((size_t)G)) || (rc = h->isconst.alloc((size_t)G)) ||
        (rc = h->colpart.alloc((size_t)h->ntiles * h->Rcol * GT)))
        return fail(rc);
    // stage the gene-major input in blocks of <= 256 genes (and <= ~256 MB)
    int64_t gc_max = std::max<int64_t>(GT, std::min<int64_t>(256, ((int64_t)1 << 28) / (n * 8) / GT * GT));
    DevBuf<double> X, D;
    if ((rc = X.alloc((size_t)gc_max * n))) return fail(rc);
    hipError_t e = hipSuccess;
    const double* cm_src = dev_x ? dev_x + dev_col0 : nullptr;  // cell-major source on the device and its row pitch
    int64_t cm_ld = dev_x ? dev_ld : G;
    if (cell_major && !dev_x) {  // one contiguous upload of vals[n][G]; the gene blocks are cut out of it on the device
        if ((rc = D.alloc((size_t)n * G))) return fail(rc);
        e = hipMemcpyAsync(D.p, vals, (size_t)n * G * 8, hipMemcpyHostToDevice, st);
        cm_src = D.p;
    }
    for (int64_t g0 = 0; g0 < G && e == hipSuccess; g0 += gc_max) {
        const int gc = (int)std::min<int64_t>(gc_max, G - g0);
        if (cm_src) {
            k_cells_to_genes<<<dim3((unsigned)ceil_div(n, GT), (unsigned)ceil_div(gc, GT)), 256, 0, st>>>(cm_src, cm_ld, n, g0, gc, X.p);
            e = hipGetLastError();
        } else {
            e = hipMemcpyAsync(X.p, vals + (size_t)g0 * n, (size_t)gc * n * 8, hipMemcpyHostToDevice, st);
        }
        if (e != hipSuccess) break;
        {
            LaunchTimer t(ctx, "autocorr_prepare");
            k_gene_stats<<<gc, 256, 0, st>>>(X.p, n, h->mean.p, h->isconst.p, g0);
            k_center_transpose<<<dim3((unsigned)ceil_div(n, GT), (unsigned)ceil_div(gc, GT)), 256, 0, st>>>(X.p, n, gc, g0, h->mean.p,
                                                                                                       h->Zt.p);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // X is reused by the next block
    }
    if (e == hipSuccess) {
        LaunchTimer t(ctx, "autocorr_spmv");
        k_rowsum<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(g->indptr.p, g->data.p, n, h->rowsum.p);
        k_spmv_tiles<<<dim3((unsigned)ceil_div(n, 4), (unsigned)h->ntiles), 256, 0, st>>>(g->indptr.p, g->indices.p, g->data.p, n,
                                                                                        h->Zt.p, h->Yt.p, h->Qt.p);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        set_error("autocorr prepare failed: %s", hipGetErrorString(e));
        return fail(SQGR_ERR_HIP);
    }
    if ((rc = column_sum(h, 1, h->Zt.p, nullptr, h->z2ss.p)) || (rc = column_sum(h, 2, h->Qt.p, nullptr, h->qsum.p))) return fail(rc);
    // W = sum of all weights (float64 accumulation, CSR order)
    std::vector<double> rs((size_t)n);
    e = hipMemcpyAsync(rs.data(), h->rowsum.p, (size_t)n * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error("autocorr prepare failed: %s", hipGetErrorString(e));
        return fail(SQGR_ERR_HIP);
    }
    double W = 0.0;
    for (int64_t i = 0; i < n; ++i) W += rs[i];
    h->W = W;
    *out = h;
    return SQGR_OK;
}

extern "C" {

int sqgr_autocorr_create(sqgr_ctx* ctx, const sqgr_graph* g, const double* vals, int64_t G, sqgr_autocorr** out) {
    return autocorr_create(ctx, g, vals, G, false, out);
}

int sqgr_autocorr_create_cm(sqgr_ctx* ctx, const sqgr_graph* g, const double* vals, int64_t G, sqgr_autocorr** out) {
    return autocorr_create(ctx, g, vals, G, true, out);
}

int sqgr_matrix_create(sqgr_ctx* ctx, const double* x, int64_t n_rows, int64_t n_cols, sqgr_matrix** out) {
    SQGR_REQUIRE(ctx && x && out && n_rows > 0 && n_cols > 0, "null argument or empty matrix");
    *out = nullptr;
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_matrix* m = new sqgr_matrix();
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    int rc = m->data.alloc((size_t)n_rows * n_cols);
    if (rc != SQGR_OK) {
        delete m;
        return rc;
    }
    hipError_t e = hipMemcpyAsync(m->data.p, x, (size_t)n_rows * n_cols * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("matrix upload failed: %s", hipGetErrorString(e));
        delete m;
        return SQGR_ERR_HIP;
    }
    *out = m;
    return SQGR_OK;
}

int sqgr_matrix_destroy(sqgr_matrix* m) {
    if (!m) return SQGR_OK;
    (void)hipSetDevice(m->ctx->device);
    delete m;
    return SQGR_OK;
}

int sqgr_autocorr_create_cols(sqgr_ctx* ctx, const sqgr_graph* g, const sqgr_matrix* m, int64_t col0, int64_t G, sqgr_autocorr** out) {
    SQGR_REQUIRE(ctx && g && m && out, "null argument");
    SQGR_REQUIRE(m->ctx == ctx, "matrix belongs to a different context");
    SQGR_REQUIRE(m->n_rows == g->n, "matrix has %lld rows, the graph %lld", (long long)m->n_rows, (long long)g->n);
    SQGR_REQUIRE(col0 >= 0 && G >= 1 && col0 + G <= m->n_cols, "columns [%lld, %lld) outside the matrix (%lld columns)", (long long)col0,
                 (long long)(col0 + G), (long long)m->n_cols);
    return autocorr_create(ctx, g, nullptr, G, true, out, m->data.p, m->n_cols, col0);
}

int sqgr_autocorr_destroy(sqgr_autocorr* h) {
    if (!h) return SQGR_OK;
    (void)hipSetDevice(h->ctx->device);
    delete h;
    return SQGR_OK;
}

int sqgr_autocorr_scores(sqgr_autocorr* h, int32_t mode, double* out_scores) {
    SQGR_REQUIRE(h && out_scores, "null argument");
    SQGR_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (moran) or 1 (geary)");
    sqgr_ctx* ctx = h->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (mode == 0)
        SQGR_TRY(column_sum(h, 0, h->Zt.p, h->Yt.p, h->tmpG.p));
    else
        SQGR_TRY(column_sum(h, 3, h->Zt.p, nullptr, h->tmpG.p));
    SQGR_TRY(h->sims.ensure((size_t)h->G));
    k_scores<<<(unsigned)ceil_div(h->G, 256), 256, 0, st>>>(mode, h->G, h->n, h->W, h->tmpG.p, h->z2ss.p, h->isconst.p, h->sims.p);
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipMemcpyAsync(out_scores, h->sims.p, (size_t)h->G * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

// permutation scores for permutations [perm_begin, perm_end): row permutations injected from the host (perm_idx), drawn
// from numpy's streams on the device (pcg_states, one row per permutation of the range) or from the device generator
static int autocorr_perms(sqgr_autocorr* h, int32_t mode, const int32_t* perm_idx, const uint64_t* pcg_states, uint64_t seed,
                          int64_t perm_begin, int64_t perm_end, double* out_sims) {
    SQGR_REQUIRE(h && out_sims, "null argument");
    SQGR_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (moran) or 1 (geary)");
    SQGR_REQUIRE(perm_begin >= 0 && perm_end >= perm_begin, "bad permutation range");
    sqgr_ctx* ctx = h->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = h->n, G = h->G, P = perm_end - perm_begin;
    if (P == 0) return SQGR_OK;
    // permutations per pass: bound idx (4 n B each) and the partials (ntiles*R*64*8 B each, x2 for Geary) to ~1 GiB each
    int R = (int)std::min<int64_t>(64, std::max<int64_t>(1, ceil_div(2048, ceil_div(std::min<int64_t>(P, 1024), PERM_TILE) * 4)));
    if (const char* env_r = getenv("SQGR_AUTOCORR_ROW_CHUNKS")) R = std::max(1, atoi(env_r));  // tuning knob
    R = (int)std::min<int64_t>(R, std::max<int64_t>(1, n / 256));
    const int64_t by_idx = std::max<int64_t>(PERM_TILE, ((int64_t)1 << 30) / (n * 4));
    const int64_t by_part = std::max<int64_t>(PERM_TILE, ((int64_t)1 << 30) / ((int64_t)h->ntiles * R * GT * 8));
    const int64_t chunk = std::min<int64_t>(std::min<int64_t>(P, 32768), std::min(by_idx, by_part));  // grid.y limit
    SQGR_TRY(h->idx.ensure((size_t)chunk * n));
    SQGR_TRY(h->part1.ensure((size_t)h->ntiles * chunk * R * GT));
    if (mode == 1) SQGR_TRY(h->part2.ensure((size_t)h->ntiles * chunk * R * GT));
    SQGR_TRY(h->sims.ensure((size_t)chunk * G));
    if (pcg_states) SQGR_TRY(h->pcg_states.ensure((size_t)chunk * 4));
    if (perm_idx)
        for (int64_t t = 0; t < P * n; ++t)
            SQGR_REQUIRE(perm_idx[t] >= 0 && perm_idx[t] < n, "perm_idx[%lld]=%d outside [0,%lld)", (long long)t, perm_idx[t], (long long)n);
    const FeistelDomain dom = make_domain((uint32_t)n);
    for (int64_t c0 = 0; c0 < P; c0 += chunk) {
        const int64_t pc = std::min(chunk, P - c0);
        if (perm_idx) {
            SQGR_HIP(hipMemcpyAsync(h->idx.p, perm_idx + (size_t)c0 * n, (size_t)pc * n * 4, hipMemcpyHostToDevice, st));
        } else if (pcg_states) {
            SQGR_HIP(hipMemcpyAsync(h->pcg_states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
            SQGR_TRY(pcg_permutations_dev(ctx, h->pcg_ws, n, h->pcg_states.p, pc, h->idx.p, st));
        } else {
            LaunchTimer t(ctx, "autocorr_perm_indices");
            k_perm_indices<<<dim3((unsigned)ceil_div(n, 256), (unsigned)pc), 256, 0, st>>>(seed, perm_begin + c0, n, dom, h->idx.p);
            SQGR_HIP(hipGetLastError());
        }
        dim3 grid((unsigned)ceil_div(pc, PERM_TILE), (unsigned)R, (unsigned)h->ntiles);
        {
            LaunchTimer t(ctx, mode == 1 ? "autocorr_perm_dot_geary" : "autocorr_perm_dot_moran");
            if (mode == 1)
                k_perm_dot<true><<<grid, 256, 0, st>>>(h->Zt.p, h->Yt.p, h->rowsum.p, h->idx.p, n, pc, R, h->part1.p, h->part2.p);
            else
                k_perm_dot<false><<<grid, 256, 0, st>>>(h->Zt.p, h->Yt.p, h->rowsum.p, h->idx.p, n, pc, R, h->part1.p, nullptr);
            SQGR_HIP(hipGetLastError());
        }
        {
            LaunchTimer t(ctx, "autocorr_perm_final");
            dim3 g2((unsigned)ceil_div(G, 256), (unsigned)pc);
            if (mode == 1)
                k_perm_final<true><<<g2, 256, 0, st>>>(h->part1.p, h->part2.p, R, pc, G, n, h->W, h->z2ss.p, h->qsum.p, h->isconst.p, h->sims.p);
            else
                k_perm_final<false><<<g2, 256, 0, st>>>(h->part1.p, nullptr, R, pc, G, n, h->W, h->z2ss.p, h->qsum.p, h->isconst.p, h->sims.p);
            SQGR_HIP(hipGetLastError());
        }
        SQGR_HIP(hipMemcpyAsync(out_sims + (size_t)c0 * G, h->sims.p, (size_t)pc * G * 8, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
    }
    return SQGR_OK;
}

int sqgr_autocorr_perms(sqgr_autocorr* h, int32_t mode, const int32_t* perm_idx, uint64_t seed, int64_t perm_begin,
                        int64_t perm_end, double* out_sims) {
    return autocorr_perms(h, mode, perm_idx, nullptr, seed, perm_begin, perm_end, out_sims);
}

int sqgr_autocorr_perms_pcg64(sqgr_autocorr* h, int32_t mode, const uint64_t* pcg_states, int64_t n_perms, double* out_sims) {
    SQGR_REQUIRE(pcg_states && n_perms >= 0, "pcg_states is NULL or n_perms < 0");
    return autocorr_perms(h, mode, nullptr, pcg_states, 0, 0, n_perms, out_sims);
}

/* the device generator's permutation indices (parity hook): int32[(perm_end-perm_begin)][n] */
int sqgr_autocorr_perm_indices(sqgr_ctx* ctx, int64_t n, uint64_t seed, int64_t perm_begin, int64_t perm_end, int32_t* out_idx) {
    SQGR_REQUIRE(ctx && out_idx && n > 0 && n < (int64_t)0x7fffffff && perm_begin >= 0 && perm_end >= perm_begin, "bad argument");
    SQGR_HIP(hipSetDevice(ctx->device));
    const int64_t P = perm_end - perm_begin;
    if (P == 0) return SQGR_OK;
    SQGR_REQUIRE(P <= 32768, "at most 32768 permutations per call");
    DevBuf<int32_t> d;
    SQGR_TRY(d.alloc((size_t)P * n));
    k_perm_indices<<<dim3((unsigned)ceil_div(n, 256), (unsigned)P), 256, 0, ctx->stream>>>(seed, perm_begin, n, make_domain((uint32_t)n), d.p);
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipMemcpyAsync(out_idx, d.p, (size_t)P * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SQGR_HIP(hipStreamSynchronize(ctx->stream));
    return SQGR_OK;
}

}  // extern "C"
