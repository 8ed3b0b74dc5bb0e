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

#include <atomic>
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

// The same from a float32 matrix (AnnData's usual `X` dtype): widened to float64 on the way — exactly what
// `.astype(float64)` does on the host, so everything downstream is bit-identical to the float64 upload.
__global__ __launch_bounds__(256) void k_cells_to_genes_f32(const float* __restrict__ D, int64_t ld, int64_t n, int64_t g0, int gc,
                                                            double* __restrict__ X) {
    __shared__ double tile[GT][GT + 1];
    const int64_t i0 = (int64_t)blockIdx.x * GT;
    const int gb = blockIdx.y * GT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < GT; r += 4) {
        const int64_t i = i0 + r;
        tile[r][tx] = (i < n && gb + tx < gc) ? (double)D[(size_t)i * ld + g0 + gb + tx] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < GT; r += 4) {
        const int64_t i = i0 + tx;
        if (gb + r < gc && i < n) X[(size_t)(gb + r) * n + i] = tile[tx][r];
    }
}

// Sparse expression (scipy CSR / CSC of the cells x genes matrix, float32 or float64 values) -> the staged gene-major block
// X[gl][i] of genes g0..g0+gc, which the caller has zeroed: the dense block `toarray()` would give, formed on the device.
// Stored entries of one cell (CSR) / one gene (CSC) land in distinct cells of X unless the matrix holds duplicates, which
// scipy sums — so does the atomic add (0 + v is exact: canonical matrices reproduce `toarray()` bit for bit).
// CSR: one thread per cell, binary search of the first stored gene >= g0 in its (sorted) row.
template <typename V>
__global__ __launch_bounds__(256) void k_csr_to_genes(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                      const V* __restrict__ values, int64_t n, int64_t g0, int gc, double* __restrict__ X) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t lo = indptr[i], hi = indptr[i + 1];
    const int64_t end = hi;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (indices[mid] < g0) lo = mid + 1; else hi = mid;
    }
    for (int64_t e = lo; e < end; ++e) {
        const int64_t c = indices[e] - g0;
        if (c >= gc) break;
        atomicAdd(&X[(size_t)c * n + i], (double)values[e]);
    }
}

// CSC: block (x, gene): the gene's stored cells, strided over the block's threads and the blocks of grid.x
template <typename V>
__global__ __launch_bounds__(256) void k_csc_to_genes(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                      const V* __restrict__ values, int64_t n, int64_t g0, double* __restrict__ X) {
    const int64_t g = blockIdx.y;
    const int64_t e0 = indptr[g0 + g], e1 = indptr[g0 + g + 1];
    for (int64_t e = e0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < e1; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&X[(size_t)g * n + indices[e]], (double)values[e]);
}

// one wave per major slice: flags[0] an index outside [0, minor), flags[1] a slice that is not ascending, flags[2] a repeated index
__global__ __launch_bounds__(256) void k_check_sparse(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t nslices,
                                                      int64_t minor, int* __restrict__ flags) {
    const int64_t j = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
    if (j >= nslices) return;
    const int64_t a = indptr[j], b = indptr[j + 1];
    int bad_range = 0, bad_order = 0, dup = 0;
    for (int64_t e = a + (threadIdx.x & 63); e < b; e += 64) {
        const int32_t c = indices[e];
        bad_range |= (c < 0 || c >= minor);
        if (e > a) {
            bad_order |= indices[e - 1] > c;
            dup |= indices[e - 1] == c;
        }
    }
    if (bad_range) flags[0] = 1;
    if (bad_order) flags[1] = 1;
    if (dup) flags[2] = 1;
}

// Feature blocks given as a LIST of columns (the reference's HVG default and explicit `genes` subsets, gr/_ppatterns.py:156-166:
// `adata[:, genes].X` — a copy of the selected columns on the host; here the selection happens on the device).
// Dense rows: the 64 x 64 LDS transpose with gathered columns.
template <typename T>
__global__ __launch_bounds__(256) void k_cells_to_genes_idx(const T* __restrict__ D, int64_t ld, int64_t n, const int32_t* __restrict__ cols, int gc,
                                                            double* __restrict__ X) {
    __shared__ double tile[GT][GT + 1];
    const int64_t i0 = (int64_t)blockIdx.x * GT;
    const int gb = blockIdx.y * GT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (gb + tx < gc) ? cols[gb + tx] : -1;
    for (int r = ty; r < GT; r += 4) {
        const int64_t i = i0 + r;
        tile[r][tx] = (i < n && c >= 0) ? (double)D[(size_t)i * ld + c] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < GT; r += 4) {
        const int64_t i = i0 + tx;
        if (gb + r < gc && i < n) X[(size_t)(gb + r) * n + i] = tile[tx][r];
    }
}

// CSC: block (x, g) expands column cols[g]
template <typename V>
__global__ __launch_bounds__(256) void k_csc_to_genes_idx(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          const V* __restrict__ values, int64_t n, const int32_t* __restrict__ cols,
                                                          double* __restrict__ X) {
    const int64_t g = blockIdx.y, c = cols[g];
    const int64_t e0 = indptr[c], e1 = indptr[c + 1];
    for (int64_t e = e0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < e1; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&X[(size_t)g * n + indices[e]], (double)values[e]);
}

// CSR -> CSC on the device (once per matrix, when a column LIST is asked for): column histogram, exclusive scan (host: n_cols + 1
// numbers), scatter through per-column cursors.  The order inside a column is whatever the atomics give — the expansion adds
// every stored entry into its own cell of a zeroed block, so it does not matter (a canonical matrix has one entry per cell).
__global__ void k_count_columns(const int32_t* __restrict__ indices, int64_t nnz, unsigned long long* __restrict__ cnt) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < nnz) atomicAdd(&cnt[indices[e]], 1ull);
}
template <typename V>
__global__ __launch_bounds__(256) void k_scatter_to_columns(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                            const V* __restrict__ values, int64_t n_rows, unsigned long long* __restrict__ cursor,
                                                            int32_t* __restrict__ out_rows, V* __restrict__ out_vals) {
    const int64_t i = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);  // one wave per row
    if (i >= n_rows) return;
    for (int64_t e = indptr[i] + (threadIdx.x & 63); e < indptr[i + 1]; e += 64) {
        const unsigned long long pos = atomicAdd(&cursor[indices[e]], 1ull);
        out_rows[pos] = (int32_t)i;
        out_vals[pos] = values[e];
    }
}

// int64 index arrays of a scipy matrix with more than 2^31 stored entries -> the library's layout
__global__ void k_narrow_indices(const int64_t* __restrict__ src, int64_t count, int32_t* __restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // an index that does not fit int32 must not wrap into the valid range: it becomes -1, which the range check that follows
    // (k_check_sparse: 0 <= index < minor) rejects like any other out-of-range index
    if (t < count) {
        const int64_t v = src[t];
        dst[t] = (v < 0 || v > (int64_t)0x7fffffff) ? -1 : (int32_t)v;
    }
}
__global__ void k_widen_indptr(const int32_t* __restrict__ src, int64_t count, int64_t* __restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < count) dst[t] = src[t];
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

// ================================================================================================================
// The LDS-bucketed permutation dot (the hot kernel for n_perms >= 512): BOTH operands of z_i * y[idx_p(i)] on chip.
//
// The spots are cut into `nch` chunks of `m` consecutive spots; a pair (i, j = idx_p(i)) of permutation p belongs to
// bucket (a, b) = (i / m, j / m).  A workgroup owns (gene pair, chunk a): Z[chunk a] of its two genes stays in LDS, the
// Y chunks b = 0 .. nch-1 pass through LDS one after the other, and lane = permutation: every lane walks the list of ITS
// permutation's pairs of bucket (a, b) and accumulates z*y in registers — two random 16-byte LDS reads per pair and two
// genes, no global gather at all.  The lists are shared by all genes (the permutations are), so their construction
// (k_bucket_*, ~1e8 pairs at config 3) is amortised over the gene block; they are stored [group of 64 permutations]
// [a][b][k][lane], padded with a pair that points at a zero row of Z to the longest list of the group (a multiple of
// LIST_UNROLL), so a wave reads 256 contiguous bytes per step and needs no tail.  Sum order: list order (ascending i)
// inside (a, b), b ascending, then a ascending in k_perm_final_lds: fixed => bit-reproducible.
constexpr int GP = 2;             // genes per LDS tile (one ds_read_b128 per operand)
constexpr int LIST_UNROLL = 8;    // pairs per lane between two waits
constexpr int LIST_ROUND = 16;    // list lengths are multiples of it (k_bucket_order schedules whole rounds of 16 classes)
constexpr int LDS_MAX_CHUNKS = 240;  // bucket kernels: (64 + 1) * nch counters in <= 64 KiB of LDS
constexpr int LDS_BYTES = 160 * 1024;
constexpr int LDS_PERM_BLOCK = 1024;  // lanes (= permutations) per workgroup
// Few permutations (the reference's everyday calls run 50-100, tests/graph/test_ppatterns.py): a workgroup of lane = permutation
// would be mostly empty.  The LDS kernel then runs on VIRTUAL permutations: permutation p is cut into LDS_SPLIT sub-lists (the
// spots i = s mod LDS_SPLIT), lane = (p, s); the kernel, the bucket lists and their layout do not change, only the list builder
// strides over the spots and k_perm_final_lds adds the LDS_SPLIT partial sums of a permutation (s ascending, then the chunks) —
// a fixed order, so a split permutation range stays bit-identical.  100 permutations fill 13 of a workgroup's 16 waves.
constexpr int LDS_SPLIT = 8;
constexpr int EXC_UNROLL = 4;         // rows of an exception list (k_perm_dot_lds RMODE 3) are multiples of it
constexpr int LIST_ROUND_SPLIT = 8;   // list lengths of the split variant (its lists are ~1/8 as long; no bank-aware order)

// The Z chunk ends in ZERO_ROWS rows of zeros, one of every bank class (row index mod 16): the padding pair of a lane that has
// no pair left points at one of them — and at any Y row — so the schedule can give it LDS banks nobody else reads at that step.
constexpr int ZERO_ROWS = 16;
// chunk length for n spots: (m + ZERO_ROWS) Z rows + m Y rows (+ m row sums for Geary) in 160 KiB
static inline int lds_chunk(int64_t n, bool geary, int* nch_out) {
    const int per_spot = 2 * GP * 8 + (geary ? 8 : 0);
    int m_max = ((LDS_BYTES - ZERO_ROWS * GP * 8 - 64) / per_spot) & ~3;  // (64 B: the row-sum class table of RMODE 2 / 3)
    if (const char* env = getenv("SQGR_AUTOCORR_LDS_CHUNK"))  // tests: several chunks on small inputs
        m_max = std::max(16, std::min(m_max, atoi(env) & ~3));
    const int64_t nch = ceil_div(n, m_max);
    *nch_out = (int)nch;
    return (int)((ceil_div(n, nch) + 3) & ~(int64_t)3);
}

// Lane -> slot of its (virtual) permutation inside a group of 64.  A `ds_read_b128` serves a wave in four fixed sets of 16 lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32; MI355X_MICROARCH.md §LDS); the slots are numbered so that set q holds the slots
// 16 q .. 16 q + 15, slot mod 16 = lane mod 16: the 16 permutations that meet in a service group — the ones whose lists are scheduled
// against each other, k_bucket_order_* — are 16 CONSECUTIVE permutations, whatever 16-aligned index a pass starts at.
__device__ __forceinline__ int perm_slot(int lane) {
    const int x = lane & 31;
    const int odd = ((x >= 4 && x < 12) || (x >= 16 && x < 20) || x >= 28) ? 1 : 0;
    return ((lane >> 5) * 2 + odd) * 16 + (lane & 15);
}

// chunk of spot j (j < 2^24: exact in float; the estimate is off by at most one)
__device__ __forceinline__ uint32_t chunk_of(uint32_t j, uint32_t m, float inv_m) {
    uint32_t b = (uint32_t)((float)j * inv_m);
    if (b * m > j) --b;
    else if ((b + 1) * m <= j) ++b;
    return b;
}

// Zp[tile2][i][c] = Zt[gene 2*tile2 + c][i]   (pair layout of the 64-gene tiles; genes past G read the tiles' zero padding)
__global__ __launch_bounds__(256) void k_repack_pairs(const double* __restrict__ Zt, int64_t n, int64_t ntiles, int64_t G2,
                                                      double* __restrict__ Zp) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // (i, gene) with gene fastest inside a 64-tile
    const int gl = (int)(t & 63);
    const int64_t i = (t >> 6) % n, tile = (t >> 6) / n;
    if (tile >= ntiles) return;
    const int64_t g = tile * GT + gl;
    if ((g >> 1) >= G2) return;
    Zp[((g >> 1) * n + i) * GP + (g & 1)] = Zt[t];
}

// len[pg][a][b] = longest list (over the 64 permutations of group pg) of bucket (a, b), rounded up to LIST_ROUND.
// grid (a, pg), one wave; lane = permutation.  Permutations >= pc have empty lists.
// S > 1: lane = virtual permutation vp = p * S + s, which owns the spots i = s (mod S) of permutation p; pc counts virtual ones.
// exc_cls != nullptr (RMODE 3, see "exception lists" below): xlen[pg][a] = the longest list of pairs whose j is not of class exc_main
__global__ __launch_bounds__(64) void k_bucket_count(const int32_t* __restrict__ idx, int64_t n, int64_t pc, int m, int nch, int S,
                                                     int round, uint32_t* __restrict__ len, const uint8_t* __restrict__ exc_cls,
                                                     int exc_main, uint32_t* __restrict__ xlen) {
    extern __shared__ uint32_t cnt[];  // [b][lane]
    const int lane = threadIdx.x, a = blockIdx.x;
    const int64_t pg = blockIdx.y, vp = pg * 64 + perm_slot(lane), p = vp / S;
    const int sub = (int)(vp - p * S);
    for (int b = 0; b < nch; ++b) cnt[b * 64 + lane] = 0;
    const float inv_m = 1.0f / (float)m;
    const int64_t i0 = (int64_t)a * m, i1 = min(n, i0 + m);
    uint32_t xc = 0;
    if (vp < pc) {
        const int32_t* row = idx + (size_t)p * n;
        if (exc_cls) {
#pragma unroll 8
            for (int64_t i = i0 + (sub - i0 % S + S) % S; i < i1; i += S) {
                const uint32_t j = (uint32_t)row[i];
                cnt[chunk_of(j, (uint32_t)m, inv_m) * 64 + lane] += 1;
                xc += exc_cls[j] != exc_main ? 1u : 0u;
            }
        } else {
#pragma unroll 8
            for (int64_t i = i0 + (sub - i0 % S + S) % S; i < i1; i += S) cnt[chunk_of((uint32_t)row[i], (uint32_t)m, inv_m) * 64 + lane] += 1;
        }
    }
    if (exc_cls) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xc = max(xc, (uint32_t)__shfl_xor((int)xc, o, 64));
        if (lane == 0) xlen[(size_t)pg * nch + a] = (xc + EXC_UNROLL - 1) / EXC_UNROLL * EXC_UNROLL;
    }
    for (int b = 0; b < nch; ++b) {
        uint32_t v = cnt[b * 64 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
        if (lane == 0) len[((size_t)pg * nch + a) * nch + b] = (v + (uint32_t)round - 1) / (uint32_t)round * (uint32_t)round;
    }
}

// off[pg][a][b] = rows (of 64 entries) in front of bucket (a, b) inside group pg's lists; total[pg] = rows of the group,
// total[npg + pg] = its longest list
__global__ __launch_bounds__(256) void k_bucket_offsets(const uint32_t* __restrict__ len, int nb, uint32_t* __restrict__ off,
                                                        uint32_t* __restrict__ total) {
    __shared__ uint32_t part[256], longest[256];
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * nb;
    const int per = (nb + 255) / 256;
    const int j0 = min(nb, tid * per), j1 = min(nb, j0 + per);
    uint32_t s = 0, mx = 0;
    for (int j = j0; j < j1; ++j) {
        s += len[base + j];
        mx = max(mx, len[base + j]);
    }
    part[tid] = s;
    longest[tid] = mx;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0, m2 = 0;
        for (int t = 0; t < 256; ++t) {
            const uint32_t v = part[t];
            part[t] = run;
            run += v;
            m2 = max(m2, longest[t]);
        }
        total[blockIdx.x] = run;
        total[gridDim.x + blockIdx.x] = m2;
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (int j = j0; j < j1; ++j) {
        off[base + j] = run;
        run += len[base + j];
    }
}

// base[pg] = rows in front of group pg (exclusive prefix of total), base[npg] = all rows, base[npg + 1] = the longest list
__global__ void k_bucket_bases(const uint32_t* __restrict__ total, int npg, uint64_t* __restrict__ base) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t run = 0;
    uint32_t mx = 0;
    for (int g = 0; g < npg; ++g) {
        base[g] = run;
        run += total[g];
        mx = max(mx, total[npg + g]);
    }
    base[npg] = run;
    base[npg + 1] = mx;
}

// lists[(base[pg] + off[pg][a][b] + k) * 64 + lane] = (i - a*m) | (idx_p(i) - b*m) << 16, pairs in ascending i; the rest of
// the bucket's len[pg][a][b] rows = the padding pair (m, 0): row m of the Z chunk is zero.
// rcls != nullptr: bits 29..31 of a pair <- the row-sum class of its j (k_perm_dot_lds RMODE 2; m < 8192)
__global__ __launch_bounds__(64) void k_bucket_fill(const int32_t* __restrict__ idx, int64_t n, int64_t pc, int m, int nch, int S,
                                                    const uint32_t* __restrict__ len, const uint32_t* __restrict__ off,
                                                    const uint64_t* __restrict__ base, uint32_t* __restrict__ lists,
                                                    const uint8_t* __restrict__ rcls, const uint8_t* __restrict__ exc_cls, int exc_main,
                                                    const uint32_t* __restrict__ xlen, const uint32_t* __restrict__ xoff,
                                                    uint32_t* __restrict__ xlists, uint16_t* __restrict__ nlen) {
    extern __shared__ uint32_t cur[];  // [b][lane], then the nch row offsets of this (pg, a)
    const int lane = threadIdx.x, a = blockIdx.x;
    const int64_t pg = blockIdx.y, vp = pg * 64 + perm_slot(lane), p = vp / S;
    const int sub = (int)(vp - p * S);
    for (int b = 0; b < nch; ++b) cur[b * 64 + lane] = 0;
    const float inv_m = 1.0f / (float)m;
    const size_t bk = ((size_t)pg * nch + a) * nch;
    uint32_t* offl = cur + nch * 64;
    for (int b = lane; b < nch; b += 64) offl[b] = off[bk + b];
    __syncthreads();
    uint32_t* out = lists + (size_t)base[pg] * 64 + lane;
    uint32_t* xout = exc_cls ? xlists + (size_t)xoff[(size_t)pg * nch + a] * 64 + lane : nullptr;
    uint32_t xk = 0;
    const int64_t i0 = (int64_t)a * m, i1 = min(n, i0 + m);
    if (vp < pc) {
        const int32_t* row = idx + (size_t)p * n;
#pragma unroll 4
        for (int64_t i = i0 + (sub - i0 % S + S) % S; i < i1; i += S) {
            const uint32_t j = (uint32_t)row[i], b = chunk_of(j, (uint32_t)m, inv_m);
            const uint32_t k = cur[b * 64 + lane];
            cur[b * 64 + lane] = k + 1;
            out[((size_t)offl[b] + k) * 64] = (uint32_t)(i - i0) | ((j - b * (uint32_t)m) << 16) | (rcls ? (uint32_t)rcls[j] << 29 : 0u);
            if (exc_cls) {
                const uint32_t c = exc_cls[j];
                if (c != (uint32_t)exc_main) {
                    xout[(size_t)xk * 64] = (uint32_t)(i - i0) | (c << 29);
                    ++xk;
                }
            }
        }
    }
    const uint32_t pad = (uint32_t)m;
    for (int b = 0; b < nch; ++b) {
        const uint32_t l = len[bk + b];
        if (nlen) nlen[(bk + b) * 64 + lane] = (uint16_t)cur[b * 64 + lane];  // (the schedule kernels choose by the lists' lengths)
        for (uint32_t k = cur[b * 64 + lane]; k < l; ++k) out[((size_t)offl[b] + k) * 64] = pad;
    }
    if (exc_cls)
        for (const uint32_t l = xlen[(size_t)pg * nch + a]; xk < l; ++xk) xout[(size_t)xk * 64] = pad;
}

// Exception lists (k_perm_dot_lds RMODE 3, Geary's C): when most spots share ONE row sum r0, sum_i z_i^2 r[idx_p(i)] is
// r0 * sum z^2 (the same for every permutation) + sum over the pairs whose j has another row sum of z_i^2 (r[j] - r0).  Those
// pairs — a few per cent on a grid: its border — get a list of their own per (group, chunk a), whatever the chunk of j: an entry
// is (i - a*m) | class of r[j] << 29, the padding pair (m, class 0) reads the zero row of Z.
// (counted and filled by k_bucket_count / k_bucket_fill in the same pass over the permutations' indices)
// xoff[.] = exclusive prefix of xlen (rows of 64 entries), xoff[count] = all rows; one workgroup
__global__ __launch_bounds__(256) void k_exc_offsets(const uint32_t* __restrict__ xlen, int count, uint32_t* __restrict__ xoff) {
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x;
    const int per = (count + 255) / 256;
    const int j0 = min(count, tid * per), j1 = min(count, j0 + per);
    uint32_t sum = 0;
    for (int j = j0; j < j1; ++j) sum += xlen[j];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 256; ++t) {
            const uint32_t v = part[t];
            part[t] = run;
            run += v;
        }
        xoff[count] = run;
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (int j = j0; j < j1; ++j) {
        xoff[j] = run;
        run += xlen[j];
    }
}

// Bank-aware order of every lane's list (any order of a list is valid; the sum order stays fixed by construction).
// A `ds_read_b128` serves a wave in 4 groups of 16 lanes, one LDS cycle per group when the 16 rows fall into 16 different
// 4-bank slots (row index mod 16); random rows collide ~3-way (tools/ubench_lds_read.hip: 11.1 instead of 4.2 clk).  Here
// lane l' of a group reads a Z row of class (k + l') mod 16 at step k: entry j of class c goes to step ((c - l') mod 16) + 16 j;
// classes with fewer entries than steps leave holes, which take the surplus entries of the fuller classes (in class order)
// and then the padding pair.  Simulated: 2.85 -> 1.46 cycles per group on the Z side (the Y side stays random).
// grid (nch * nch buckets, groups); one wave; lists longer than ORDER_MAX rows are left as they are.
constexpr int ORDER_MAX = 320;
// The rotation l' = (global permutation index) mod 16 — every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...,
// MI355X_MICROARCH.md §LDS) holds each residue once — and the schedule runs over the lane's OWN length rounded up to 16, so the
// order of a permutation's list depends on that permutation alone: results stay bit-identical however a range is split.
// cls_shift 0: classes of the Z row (i mod 16); 16: classes of the Y row instead (SQGR_AUTOCORR_ORDER_BY=y, an experiment of round 4:
// the Y side then runs conflict-free and the Z side random — measured the same to within noise).
__global__ __launch_bounds__(64) void k_bucket_order(int m, int nb, int64_t perm0, const uint32_t* __restrict__ len,
                                                     const uint32_t* __restrict__ off, const uint64_t* __restrict__ base,
                                                     uint32_t* __restrict__ lists, int cls_shift) {
    __shared__ uint32_t ent[ORDER_MAX * 64];   // [k][lane] the list as built (ascending i)
    __shared__ uint16_t hole[ORDER_MAX * 64];  // [s][lane] step of the lane's s-th hole
    __shared__ uint16_t ncls[16 * 64], rank[16 * 64], sbase[16 * 64];  // per lane and class: entries, running rank, surplus offset
    const int lane = threadIdx.x;
    const int64_t pg = blockIdx.y;
    const size_t bk = (size_t)pg * nb + blockIdx.x;
    const int L = (int)len[bk];
    if (L == 0 || L > ORDER_MAX) return;
    uint32_t* lst = lists + ((size_t)base[pg] + off[bk]) * 64 + lane;
    const uint32_t pad = (uint32_t)m;
    const int lp = (int)((perm0 + pg * 64 + lane) & 15);
    for (int c = 0; c < 16; ++c) ncls[c * 64 + lane] = rank[c * 64 + lane] = 0;
    int cnt = 0;
    for (int k0 = 0; k0 < L; k0 += 16) {  // L is a multiple of LIST_ROUND = 16: sixteen loads in flight per lane
        uint32_t e[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) e[u] = lst[(size_t)(k0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            ent[(k0 + u) * 64 + lane] = e[u];
            if (e[u] != pad) {
                ncls[((e[u] >> cls_shift) & 15u) * 64 + lane] += 1;
                ++cnt;
            }
        }
    }
    const int Lo = (cnt + 15) & ~15, D = Lo >> 4;  // own length: D demands of every class
    int surplus = 0;
    for (int c = 0; c < 16; ++c) {
        sbase[c * 64 + lane] = (uint16_t)surplus;
        surplus += max(0, (int)ncls[c * 64 + lane] - D);
    }
    int nholes = 0;
    for (int k = 0; k < L; ++k) {  // holes in step order; they hold the padding pair unless a surplus entry lands there
        const int c = (k + lp) & 15;
        if (k >= Lo || (k >> 4) >= (int)ncls[c * 64 + lane]) {
            if (k < Lo) {
                hole[nholes * 64 + lane] = (uint16_t)k;
                ++nholes;
            }
            lst[(size_t)k * 64] = pad;
        }
    }
    for (int k = 0; k < L; ++k) {
        const uint32_t e = ent[k * 64 + lane];
        if (e == pad) continue;
        const int c = (int)((e >> cls_shift) & 15u);
        const int j = rank[c * 64 + lane];
        rank[c * 64 + lane] = (uint16_t)(j + 1);
        const int pos = j < D ? ((c - lp) & 15) + 16 * j : (int)hole[((int)sbase[c * 64 + lane] + j - D) * 64 + lane];
        lst[(size_t)pos * 64] = e;
    }
}

// The JOINT schedule (round 4, the default): the order above makes one side of a pair conflict-free and leaves the other random
// (1.46 + 2.84 LDS cycles per 16-lane group and pair).  A list may be walked in ANY order, and the 16 permutations that share a
// `ds_read_b128` lane group can agree on theirs: lane l' still reads a Z row of class (k + l') mod 16 at step k, and among its pairs
// of that class it takes one whose Y row class no other lane of the group has taken at that step.  Simulated 1.32 + 1.67 cycles.
//   cell (zc, yc) = (i mod 16, j mod 16) of a pair; pool (lane, zc) = the lane's pairs with Z class zc.
//   main phase: thread (group q, offset o) owns the steps k = o + 16 j and the pools (lane, (o + l') mod 16) of its group — no
//     other thread touches them.  Per step it visits the 16 lanes, smallest pool first (fixed order), and gives each a free Y
//     class of its pool (rotating start) or, failing that, any.
//   surplus phase: pools fuller than the lane has rounds keep pairs, emptier ones leave holes; every lane fills its holes with the
//     remaining pairs that collide least with what the main phase gave the other lanes at that step (Y first, then Z).
// The schedule of a lane depends on the 15 lanes it shares the group with: the caller builds lists for whole 16-aligned groups of
// the GLOBAL permutation index (perm_slot; sqgr_autocorr_perms generates the missing neighbours), so a split range stays bit-identical.
// Lists longer than ORDER_SEG rows are scheduled in independent segments of ORDER_SEG rows (grid.z).  Out of place: src -> dst.
constexpr int ORDER_SEG = 256;

template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v) {  // the value of the lane N places along the 16-lane row (rotating)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 | N, 0xf, 0xf, false);
}
// lane of `ds_read_b128` group q that holds residue r = lane mod 16 (groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32)
__device__ __forceinline__ int group_lane(int q, int r) {
    const bool mid = r >= 4 && r < 12;
    return (q >> 1) * 32 + (((q & 1) != 0) == mid ? r : 16 + r);
}

// grid (buckets * 4 lane groups, groups of 64 permutations, segments); 16 threads: the schedule is a chain of dependent LDS accesses,
// what counts is how many groups a CU holds at once (15 KiB of LDS each)
__global__ __launch_bounds__(16) void k_bucket_order_joint(int m, int nb, const uint32_t* __restrict__ len, const uint32_t* __restrict__ off,
                                                           const uint64_t* __restrict__ base, const uint32_t* __restrict__ src,
                                                           uint32_t* __restrict__ dst, const uint16_t* __restrict__ nlen, int longer_than) {
    __shared__ uint16_t cw[256 * 16];    // [cell][r] pairs of the cell not yet placed << 8 | cursor into stp (mod 256; cells ascending)
    __shared__ uint8_t stp[256 * 16];    // [.][r] the steps given to the lane's pairs, grouped by cell
    __shared__ uint16_t amask[16 * 16];  // [zc][r] Y classes pool (lane r, zc) still holds
    __shared__ uint16_t psize[16 * 16];  // [zc][r] pairs in pool (lane r, zc)
    __shared__ uint16_t hbits[16 * 16];  // [o][r] bit j: step o + 16 j of lane r is taken
    __shared__ uint32_t yz[ORDER_SEG];   // [k] Y classes (bits 0..15) and Z classes (16..31) the group reads at step k
    __shared__ uint16_t Dl[16];          // rounds of lane r (its own length / 16, rounded up)
    __shared__ int saturated;
    const int t = threadIdx.x, q = (int)(blockIdx.x & 3u);
    const int lane = group_lane(q, t);
    const int64_t pg = blockIdx.y;
    const size_t bk = (size_t)pg * nb + (blockIdx.x >> 2);
    const int L = (int)len[bk], s0 = (int)blockIdx.z * ORDER_SEG;
    if (s0 >= L) return;
    if (longer_than > 0) {  // only the lane groups whose longest list exceeds `longer_than` rows (the others: k_bucket_order_steps)
        uint32_t h = nlen[bk * 64 + lane];
        h = max(h, row_ror<1>(h));
        h = max(h, row_ror<2>(h));
        h = max(h, row_ror<4>(h));
        if ((int)max(h, row_ror<8>(h)) <= longer_than) return;
    }
    const int Ls = min(ORDER_SEG, L - s0);  // a multiple of 16
    const size_t row0 = ((size_t)base[pg] + off[bk] + (size_t)s0) * 64;
    const uint32_t* a = src + row0 + lane;
    uint32_t* d = dst + row0 + lane;
    const uint32_t pad = (uint32_t)m;
    for (int i = t; i < 256 * 8; i += 16) reinterpret_cast<uint32_t*>(cw)[i] = 0;
    for (int i = t; i < ORDER_SEG; i += 16) yz[i] = 0;
    if (t == 0) saturated = 0;
    __syncthreads();
    // --- thread = lane r: the cells of its list
    int cnt = 0;
    bool sat = false;
    for (int k0 = 0; k0 < Ls; k0 += 16) {
        uint32_t e[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) e[u] = a[(size_t)(k0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if ((e[u] & 0xffffu) < pad) {
                const int cell = ((((e[u] & 15u) << 4) | ((e[u] >> 16) & 15u))) * 16 + t;
                const uint32_t w = cw[cell];
                sat |= w >= 0xff00u;
                cw[cell] = (uint16_t)(w + 0x100u);
                ++cnt;
            }
    }
    if (sat) saturated = 1;
    __syncthreads();
    if (saturated) {  // 256 pairs of one lane in one cell (an 8-bit counter): such a list has one order
        for (int k = 0; k < Ls; ++k) d[(size_t)k * 64] = a[(size_t)k * 64];
        return;
    }
    Dl[t] = (uint16_t)((cnt + 15) >> 4);
    {
        int run = 0;
        for (int zc = 0; zc < 16; ++zc) {
            uint32_t mask = 0;
            int ps = 0;
#pragma unroll
            for (int y = 0; y < 16; ++y) {
                const int c = cw[(zc * 16 + y) * 16 + t] >> 8;
                cw[(zc * 16 + y) * 16 + t] = (uint16_t)((c << 8) | (run & 255));
                run += c;
                ps += c;
                mask |= c ? (1u << y) : 0u;
            }
            amask[zc * 16 + t] = (uint16_t)mask;
            psize[zc * 16 + t] = (uint16_t)ps;
        }
    }
    __syncthreads();
    // --- thread = offset o: the steps k = o + 16 j and the pools (lane r, (o + r) mod 16)
    {
        const int o = t;
        int Dmax = 0;
        uint64_t visit = 0;  // residues in visiting order, 4 bits each: smallest pool first
        {
            uint32_t key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                key[r] = ((uint32_t)psize[((o + r) & 15) * 16 + r] << 4) | (uint32_t)r;
                Dmax = max(Dmax, (int)Dl[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rank = 0;
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) rank += key[r2] < key[r] ? 1 : 0;
                visit |= (uint64_t)r << (4 * rank);
            }
        }
        uint32_t taken[16];  // per lane: bit j = step o + 16 j given away
#pragma unroll
        for (int r = 0; r < 16; ++r) taken[r] = 0;
        for (int j = 0; j < Dmax; ++j) {
            const int k = o + 16 * j;
            uint32_t used = 0, zused = 0, got = 0;
            for (int i = 0; i < 16; ++i) {
                const int r = (int)((visit >> (4 * i)) & 15u);
                if (j >= (int)Dl[r]) continue;
                const int zc = (o + r) & 15;
                const uint32_t avail = amask[zc * 16 + r];
                if (!avail) continue;
                uint32_t pick = avail & ~used;
                if (!pick) pick = avail;
                const int s = (j * 5 + o * 3 + r * 7) & 15;
                const uint32_t rot = ((pick >> s) | (pick << (16 - s))) & 0xffffu;
                const int y = (__builtin_ctz(rot) + s) & 15;
                const int cell = (zc * 16 + y) * 16 + r;
                const uint32_t w = cw[cell];
                cw[cell] = (uint16_t)(((w - 0x100u) & 0xff00u) | ((w + 1u) & 0xffu));
                if ((w >> 8) == 1u) amask[zc * 16 + r] = (uint16_t)(avail & ~(1u << y));
                stp[(w & 0xffu) * 16 + r] = (uint8_t)k;
                used |= 1u << y;
                zused |= 1u << zc;
                got |= 1u << r;
            }
            yz[k] = used | (zused << 16);
#pragma unroll
            for (int r = 0; r < 16; ++r) taken[r] |= ((got >> r) & 1u) << j;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) hbits[o * 16 + r] = (uint16_t)taken[r];
    }
    __syncthreads();
    // --- thread = lane r: the pairs left over go to the lane's holes, judged by the classes the OTHER lanes read at a hole's step (as
    // the main phase left them: the lanes do this side by side).  Three sweeps over the holes: a pair that collides on neither
    // side, then one whose Y row does not collide, then any.
    {
        const uint32_t dmask = (1u << Dl[t]) - 1u;
        uint32_t zrem = 0;
        int rem = cnt;
        for (int c = 0; c < 16; ++c) {
            zrem |= amask[c * 16 + t] ? (1u << c) : 0u;
            rem -= __builtin_popcount(hbits[c * 16 + t]);
        }
        for (int lvl = 0; lvl < 3; ++lvl)
            for (int o2 = 0; o2 < 16; ++o2) {
                uint32_t hb = hbits[o2 * 16 + t];
                for (uint32_t hh = rem ? (~hb & dmask) : 0u; hh && rem; hh &= hh - 1) {
                    const int j = __builtin_ctz(hh), k = o2 + 16 * j;
                    const uint32_t w = yz[k];
                    for (uint32_t zcand = lvl == 0 ? (zrem & ~(w >> 16)) : zrem; zcand; zcand &= zcand - 1) {
                        const int zc = __builtin_ctz(zcand);
                        uint32_t am = amask[zc * 16 + t];
                        const uint32_t ym = (lvl == 2 ? am : (am & ~w)) & 0xffffu;
                        if (!ym) continue;
                        const int y = __builtin_ctz(ym);
                        const int cell = (zc * 16 + y) * 16 + t;
                        const uint32_t v = cw[cell];
                        cw[cell] = (uint16_t)(((v - 0x100u) & 0xff00u) | ((v + 1u) & 0xffu));
                        if ((v >> 8) == 1u) {
                            am &= ~(1u << y);
                            amask[zc * 16 + t] = (uint16_t)am;
                            if (!am) zrem &= ~(1u << zc);
                        }
                        stp[(v & 0xffu) * 16 + t] = (uint8_t)k;
                        hb |= 1u << j;
                        --rem;
                        break;
                    }
                }
                hbits[o2 * 16 + t] = (uint16_t)hb;
            }
    }
    // --- (same thread) every pair to its step — the cursors now stand at the end of their cells —, the padding pair elsewhere
    // (a padding pair: the zero row of the lane's own Z class at that step, a Y row of a class the main phase left free there)
    for (int k = 0; k < Ls; ++k)
        if (!((hbits[(k & 15) * 16 + t] >> (k >> 4)) & 1u)) {
            const uint32_t yfree = ~yz[k] & 0xffffu;
            const uint32_t rot = ((yfree >> t) | (yfree << (16 - t))) & 0xffffu;
            const uint32_t yc = yfree ? (uint32_t)((__builtin_ctz(rot) + t) & 15) : 0u;
            d[(size_t)k * 64] = (pad + (((uint32_t)(k + t) - pad) & 15u)) | (yc << 16);
        }
    for (int k0 = 0; k0 < Ls; k0 += 16) {
        uint32_t e[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) e[u] = a[(size_t)(k0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if ((e[u] & 0xffffu) < pad) {
                const int cell = ((((e[u] & 15u) << 4) | ((e[u] >> 16) & 15u))) * 16 + t;
                const uint32_t v = (uint32_t)cw[cell] - 1u;
                cw[cell] = (uint16_t)v;  // (only the cursor byte is read from here on)
                d[(size_t)stp[(v & 0xffu) * 16 + t] * 64] = e[u];
            }
    }
}

// The STEP schedule (the default): the rotation above gives every Z class the same number of slots in a lane's list, so the ~9 % of
// a lane's pairs that exceed their class's share land in other classes' holes and collide there.  A list of row classes can be cut
// into steps in which all 16 lanes read DIFFERENT Z classes whatever the counts (an edge colouring of the lanes x classes
// multigraph: as many steps as the longest lane or the fullest class needs, and the group's longest list leaves that slack).
// Step by step, for one lane group:
//   Z side, every four steps: the classes, fullest first (most pairs left over all lanes), each take the lane with the most pairs
//     left among the lanes that hold the class and have no class yet; a lane keeps its class for the four steps (idle if the pool
//     runs dry: simulated no worse, the Y side gains from drawing on one pool); a lane that can no longer afford to idle takes any
//     class it holds.
//   Y side, every step: the 16 Y classes in turn (rotating start) each take, among the lanes whose pool (lane, its Z class) holds
//     the class, the one with the fewest alternatives; a lane no class took reads a colliding one.
//   An idle lane reads a padding pair: the zero row (ZERO_ROWS) of a Z class and a Y row of a class nobody reads at that step.
// Simulated 1.02 + 1.34 cycles per step and group (the rotation: 1.39 + 1.67; random rows: 2.85 + 2.85).  "Class c takes the best
// lane that holds it" is integer arithmetic on a DPP row (one lane group = 16 threads): the lanes' holdings move to their places in
// order of preference (ds_permute), 16 wave ballots give every class's holders in that order, the turns are scalar arithmetic
// (lowest set bit = best holder still free), the lanes read their class back (ds_bpermute).  A workgroup is two lane groups,
// 32 KiB of LDS, 5 per CU; the kernel is bound by instruction issue (PMC: 35 % VALU, 24 % SALU of its wave cycles).
constexpr int STEP_SEG = 384;  // rows scheduled at a time (longer lists: independent segments, the full ones without slack)

__device__ __forceinline__ uint32_t row_or(uint32_t v) {
    v |= row_ror<1>(v);
    v |= row_ror<2>(v);
    v |= row_ror<4>(v);
    return v | row_ror<8>(v);
}
// position of the n-th (0-based) set bit of a 16-bit mask (n < popcount)
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t mask, uint32_t n) {
    uint32_t pos = 0, c = (uint32_t)__builtin_popcount(mask & 0xffu);
    if (n >= c) { n -= c; pos = 8; }
    c = (uint32_t)__builtin_popcount((mask >> pos) & 0xfu);
    if (n >= c) { n -= c; pos += 4; }
    c = (uint32_t)__builtin_popcount((mask >> pos) & 0x3u);
    if (n >= c) { n -= c; pos += 2; }
    return pos + (n >= ((mask >> pos) & 1u) ? 1u : 0u);
}
// the lanes of the own row whose (distinct) key is larger, as a mask over the wave's lanes; src[N] = 1 << (lane row_ror<N> reads)
#define SQGR_ROW_OTHERS(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
struct RowSources {
    uint32_t bit[16];
};
__device__ __forceinline__ uint32_t row_higher(uint32_t key, const RowSources& src) {
    uint32_t hp = 0;
#define SQGR_HP(N) hp |= row_ror<N>(key) > key ? src.bit[N] : 0u;
    SQGR_ROW_OTHERS(SQGR_HP)
#undef SQGR_HP
    return hp;
}

// grid (buckets, groups of 64 permutations); 64 threads = a whole wavefront = the four lane groups of a bucket's 64 lists (round 5:
// 32 threads — two lane groups, half of every wave instruction masked off — until then; the step array, 432 of a thread's 1010
// bytes of LDS, now lives in a global scratch array shaped like the lists: fire-and-forget stores during the turns, read back by
// the last pass — 36 KiB per workgroup, four per CU = 16 lane groups in flight instead of 10).
// "Class c takes the best lane that holds it" = one wave ballot of the holders + each lane's mask of the better lanes of its row.
__global__ __launch_bounds__(64) void k_bucket_order_steps(int m, int nb, const uint32_t* __restrict__ len, const uint32_t* __restrict__ off,
                                                           const uint64_t* __restrict__ base, const uint32_t* __restrict__ src,
                                                           uint32_t* __restrict__ dst, const uint16_t* __restrict__ nlen,
                                                           uint16_t* __restrict__ stp_scr) {
    __shared__ uint8_t cnt[256 * 64];            // [cell][t] pairs of the cell not yet placed
    __shared__ uint8_t cur[256 * 64];            // [cell][t] slot of the cell's next pair, counted from the start of its pool
    __shared__ uint16_t pstart[17 * 64];         // [zc][t] pairs in the pools below zc; [16]: all
    __shared__ uint16_t amask[16 * 64];          // [zc][t] Y classes pool (lane, zc) still holds
    __shared__ int saturated[4];                 // per lane group
    const int t = threadIdx.x, r = t & 15;
    const int lane = group_lane(t >> 4, r);
    const int64_t pg = blockIdx.y;
    const size_t bk = (size_t)pg * nb + blockIdx.x;
    const int L = (int)len[bk];
    if (L == 0) return;
    // A lane group whose longest list fits STEP_SEG rows is scheduled here, in one piece; the others keep the rotation schedule in
    // segments (k_bucket_order_joint): cut into full segments the step schedule has no slack to work with (measured at 30 000 spots,
    // lists of ~830 pairs: 18.5 vs 17.2 ms).  The choice looks at the group's own lists only (split invariance).
    uint32_t horizon = nlen[bk * 64 + lane];
    horizon = max(horizon, row_ror<1>(horizon));
    horizon = max(horizon, row_ror<2>(horizon));
    horizon = max(horizon, row_ror<4>(horizon));
    horizon = (max(horizon, row_ror<8>(horizon)) + 15u) & ~15u;  // the group's longest list, rounded up to 16: its steps
    const bool mine = horizon <= (uint32_t)STEP_SEG;
    if (!__builtin_amdgcn_ballot_w64(mine)) return;
    const int Ls = min(STEP_SEG, L);  // a multiple of 16
    const size_t row0 = ((size_t)base[pg] + off[bk]) * 64;
    const uint32_t* a = src + row0 + lane;
    uint32_t* d = dst + row0 + lane;
    uint16_t* stp = stp_scr + row0 + lane;       // [slot * 64]: the step given to the lane's pair of that slot (slots < the lane's pairs <= L)
    const uint32_t pad = (uint32_t)m;
    for (int i = t; i < 256 * 16; i += 64) reinterpret_cast<uint32_t*>(cnt)[i] = 0;
    if (t < 4) saturated[t] = 0;
    __syncthreads();
    // --- thread = lane: the cells of its list
    uint32_t rem = 0;
    bool sat = false;
    for (int k0 = 0; k0 < Ls; k0 += 16) {
        uint32_t e[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) e[u] = a[(size_t)(k0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (mine && (e[u] & 0xffffu) < pad) {
                const int cell = ((((e[u] & 15u) << 4) | ((e[u] >> 16) & 15u))) * 64 + t;
                const uint32_t c = cnt[cell];
                sat |= c == 255u;
                cnt[cell] = (uint8_t)(c + 1u);
                ++rem;
            }
    }
    uint32_t cand = 0;  // Z classes the lane still holds pairs of
    {
        uint32_t run = 0;
        for (int zc = 0; zc < 16; ++zc) {
            uint32_t mask = 0, ps = 0;
#pragma unroll
            for (int y = 0; y < 16; ++y) {
                const uint32_t c = cnt[(zc * 16 + y) * 64 + t];
                cur[(zc * 16 + y) * 64 + t] = (uint8_t)ps;
                ps += c;
                mask |= c ? (1u << y) : 0u;
            }
            sat |= ps > 255u;
            amask[zc * 64 + t] = (uint16_t)mask;
            pstart[zc * 64 + t] = (uint16_t)run;
            run += ps;
            cand |= mask ? (1u << zc) : 0u;
        }
        pstart[16 * 64 + t] = (uint16_t)run;
    }
    if (sat) saturated[t >> 4] = 1;
    __syncthreads();
    bool mine_sched = mine;
    if (mine && saturated[t >> 4]) {  // 256 pairs of one lane in one Z class (8-bit counters): the group's lists stay as built
        for (int k = 0; k < L; ++k) d[(size_t)k * 64] = a[(size_t)k * 64];
        mine_sched = false;
        rem = 0;
        cand = 0;
    }
    if (!__builtin_amdgcn_ballot_w64(mine_sched)) return;
    uint32_t cdeg = 0;  // thread r of a row also keeps class r: the pairs of that class all 16 lanes still hold
#pragma unroll
    for (int l2 = 0; l2 < 16; ++l2) cdeg += (uint32_t)pstart[(r + 1) * 64 + (t & 48) + l2] - (uint32_t)pstart[r * 64 + (t & 48) + l2];
    RowSources rs;
    rs.bit[0] = 0;
#define SQGR_SRC(N) rs.bit[N] = 1u << (row_ror<N>((uint32_t)t) & 31u);  // (one bit per lane of the own row: distinct within a row)
    SQGR_ROW_OTHERS(SQGR_SRC)
#undef SQGR_SRC
    uint32_t sig_lo = 0, sig_hi = 0;  // the classes, fullest first, 4 bits each
    uint32_t zpos = 0;                // (byte address of) the lane's place in its row by pairs left, most first
    const uint32_t rowbase = (uint32_t)(t & 48);
    // "The classes in turn each take the best lane that holds them": the lanes' holdings are moved to their places in order of
    // preference (ds_permute), 16 wave ballots give every class's holders in that order, the turns are scalar arithmetic on the
    // masks (lowest set bit = best holder still free, one 16-bit half per lane group), and the lanes read their class back.
    auto take_turns = [&](const uint64_t (&holders)[16], uint64_t (&won)[16]) {
        uint64_t free_places = ~0ull;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint64_t h = holders[i] & free_places;
            const uint64_t f0 = h & 0xffffull, f1 = h & 0xffff0000ull, f2 = h & 0xffff00000000ull, f3 = h & 0xffff000000000000ull;
            won[i] = (f0 & (0ull - f0)) | (f1 & (0ull - f1)) | (f2 & (0ull - f2)) | (f3 & (0ull - f3));
            free_places &= ~won[i];
        }
    };
    uint32_t zkeep = 16;  // the lane's Z class of the current four steps
    // The steps of a lane group: its own longest list — NOT the bucket's rows (the longest of all 64 lanes): the schedule of a group
    // must not depend on the other groups (split invariance); the rows past it hold padding pairs.
    const uint32_t hz = mine_sched ? horizon : 0u;
    const int steps = (int)max(max((uint32_t)__builtin_amdgcn_readlane((int)hz, 0), (uint32_t)__builtin_amdgcn_readlane((int)hz, 16)),
                               max((uint32_t)__builtin_amdgcn_readlane((int)hz, 32), (uint32_t)__builtin_amdgcn_readlane((int)hz, 48)));
    if (mine_sched)
        for (int k = (int)horizon; k < L; ++k) d[(size_t)k * 64] = (pad + (((uint32_t)r - pad) & 15u)) | ((uint32_t)r << 16);
    for (int k = 0; k < steps; ++k) {
        if ((k & 3) == 0) {
            // Z side, every FOUR steps (a lane whose pool runs dry in between idles until the next turn: simulated no worse — the Y side
            // even gains from drawing on one pool — and the lane groups read distinct classes all the same): the classes, fullest
            // first, each take the lane with the most pairs left among their holders without a class
            const uint32_t ck = (cdeg << 4) | (uint32_t)(15 - r);
            const uint32_t rank = (uint32_t)__builtin_popcount(row_higher(ck, rs));
            sig_lo = row_or(rank < 8 ? (uint32_t)r << (4 * rank) : 0u);
            sig_hi = row_or(rank >= 8 ? (uint32_t)r << (4 * (rank - 8)) : 0u);
            zpos = (rowbase + (uint32_t)__builtin_popcount(row_higher((rem << 4) | (uint32_t)(15 - r), rs))) << 2;
            const uint32_t held = (uint32_t)__builtin_amdgcn_ds_permute((int)zpos, (int)(rem ? cand : 0u));  // (by place)
            uint32_t zi[16];
            uint64_t holders[16], won[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                zi[i] = ((i < 8 ? sig_lo : sig_hi) >> (4 * (i & 7))) & 15u;
                holders[i] = __builtin_amdgcn_ballot_w64(((held >> zi[i]) & 1u) != 0);
            }
            take_turns(holders, won);
            uint32_t zp = 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) zp = __builtin_amdgcn_inverse_ballot_w64(won[i]) ? zi[i] : zp;
            zkeep = (uint32_t)__builtin_amdgcn_ds_bpermute((int)zpos, (int)zp);
        }
        uint32_t myz = (rem && zkeep < 16 && ((cand >> zkeep) & 1u)) ? zkeep : 16u;
        if (rem && myz == 16 && rem + (uint32_t)k >= horizon) myz = (uint32_t)__builtin_ctz(cand);  // no step to spare: a class some lane may read
        const bool active = myz < 16;
        const uint32_t zused = row_or(active ? (1u << myz) : 0u);
        cdeg -= ((zused >> r) & 1u) && cdeg ? 1u : 0u;
        // Y side: the classes in turn (rotating start) each take the lane with the fewest alternatives among the lanes whose pool holds them
        const uint32_t av = active ? (uint32_t)amask[myz * 64 + t] : 0u;
        uint32_t myy = 16;
        {
            const uint32_t ykey = active ? (((17u - (uint32_t)__builtin_popcount(av)) << 4) | (uint32_t)(15 - r)) : (uint32_t)(15 - r);
            const uint32_t ypos = (rowbase + (uint32_t)__builtin_popcount(row_higher(ykey, rs))) << 2;
            const uint32_t held = (uint32_t)__builtin_amdgcn_ds_permute((int)ypos, (int)av);
            const uint32_t y0 = (uint32_t)(k * 5);
            uint64_t holders[16], won[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) holders[i] = __builtin_amdgcn_ballot_w64(((held >> ((y0 + (uint32_t)i) & 15u)) & 1u) != 0);
            take_turns(holders, won);
            uint32_t yp = 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) yp = __builtin_amdgcn_inverse_ballot_w64(won[i]) ? ((y0 + (uint32_t)i) & 15u) : yp;
            myy = (uint32_t)__builtin_amdgcn_ds_bpermute((int)ypos, (int)yp);
        }
        const bool unas = active && myy == 16;
        if (active && unas) myy = (uint32_t)__builtin_ctz(av);  // every class of the pool is taken: collide
        // the idle lanes read padding pairs: the zero row of a Z class and a Y row of a class no lane reads at this step
        const uint32_t idle = row_or(active ? 0u : (1u << r));
        const uint32_t yused = row_or(active ? (1u << myy) : 0u);
        if (active) {
            const int cell = (int)(myz * 16 + myy) * 64 + t;
            const uint32_t c = cnt[cell], slot = (uint32_t)pstart[myz * 64 + t] + cur[cell];
            cnt[cell] = (uint8_t)(c - 1u);
            cur[cell] = (uint8_t)(cur[cell] + 1u);
            if (c == 1u) {
                const uint32_t nav = av & ~(1u << myy);
                amask[myz * 64 + t] = (uint16_t)nav;
                if (!nav) cand &= ~(1u << myz);
            }
            stp[(size_t)slot * 64] = (uint16_t)k;
            --rem;
        } else if (mine_sched && (uint32_t)k < horizon) {
            const uint32_t nth = (uint32_t)__builtin_popcount(idle & ((1u << r) - 1u));  // (at least as many free classes as idle lanes)
            const uint32_t zc = nth_set_bit(~zused & 0xffffu, nth), yc = nth_set_bit(~yused & 0xffffu, nth);
            d[(size_t)k * 64] = (pad + ((zc - pad) & 15u)) | (yc << 16);
        }
    }
    __threadfence_block();  // the step array is read back by the lane that wrote it
    __syncthreads();
    // --- every pair to its step: the cursors now stand at the end of their cells
    for (int k0 = 0; k0 < Ls; k0 += 16) {
        uint32_t e[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) e[u] = a[(size_t)(k0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (mine_sched && (e[u] & 0xffffu) < pad) {
                const uint32_t zc = e[u] & 15u;
                const int cell = (int)((zc << 4) | ((e[u] >> 16) & 15u)) * 64 + t;
                const uint32_t within = (uint32_t)cur[cell] - 1u;
                cur[cell] = (uint8_t)within;
                const uint32_t slot = (uint32_t)pstart[zc * 64 + t] + (within & 0xffu);
                const uint32_t k = (uint32_t)stp[(size_t)slot * 64];
                d[(size_t)k * 64] = e[u];
            }
    }
}
#undef SQGR_ROW_OTHERS

// grid (ceil(G2 / 256) * 256 * nch, perm blocks); block = 64 * (groups in a perm block) lanes, lane = permutation.
// Block -> (gene pair, chunk a) with XCD affinity: hardware places block b on XCD (b % 8); the 32 workgroups an XCD runs
// side by side own 32 different gene pairs and the SAME chunk a, so they walk the same lists in step and the lists
// (shared by all genes) are served by that XCD's L2 — only the Y chunks are private traffic.
// part1[((tile2 * pc + p) * nch + a) * 2 + c]
constexpr int LDS_STAGE = 5;  // rows per thread of one chunk at LDS_PERM_BLOCK threads (m <= 5 * 1024)
// RMODE 0: Moran's I.  Geary's C also needs sum_i z_i^2 r[idx_p(i)], r = the graph's row sums: RMODE 1 keeps r[chunk b] in LDS next to
// Y (a third random read per pair, 4092-spot chunks); RMODE 2 (round 4): the row sums take at most 8 DISTINCT values — a
// row-normalised graph has one per degree, a binary graph its degrees — so the list entry carries the class of r[j] in its top three
// bits and the kernel reads an 8-entry table: Moran's chunks, a read that meets in at most 8 addresses.
template <int RMODE>
__global__ __launch_bounds__(LDS_PERM_BLOCK) void k_perm_dot_lds(const double* __restrict__ Zp, const double* __restrict__ Yp,
                                                                 const double* __restrict__ rowsum, int64_t n, int64_t pc, int npg,
                                                                 int64_t G2, int m, int nch, const uint32_t* __restrict__ len,
                                                                 const uint32_t* __restrict__ off, const uint64_t* __restrict__ base,
                                                                 const uint32_t* __restrict__ lists, double* __restrict__ part1,
                                                                 double* __restrict__ part2, int a_per_xcd,
                                                                 const uint32_t* __restrict__ xlen, const uint32_t* __restrict__ xoff,
                                                                 const uint32_t* __restrict__ xlists) {
    // RMODE 3: the main loop is Moran's; the z^2 r term comes from the exception lists (see k_exc_offsets) and a constant
    constexpr bool SECOND = RMODE != 0, GEARY = RMODE == 1 || RMODE == 2, RARR = RMODE == 1, RCLS = RMODE == 2, REXC = RMODE == 3;
    // pairs per lane between two waits (register budget: 128; the class-table variant spills 6 registers at 8 and is still 8 % faster
    // than at 4: 70.8 vs 77.2 ms per 2048 genes x 1000 permutations)
    constexpr int UNR = RARR ? LIST_UNROLL / 2 : LIST_UNROLL;
    constexpr uint32_t JMASK = RCLS ? 0x1fffu : 0xffffu;        // RCLS: bits 29..31 of an entry hold the class of r[j]
    extern __shared__ double2 smem2[];
    double2* Zc = smem2;            // [m + ZERO_ROWS] rows (z of gene 0, z of gene 1); rows m .. = 0
    double2* Yc = smem2 + (m + ZERO_ROWS);  // [m]
    double* Rc = reinterpret_cast<double*>(Yc + m);  // RMODE 1: [m] row sums; RMODE 2: the table of the (<= 8) distinct row sums
    const int tid = threadIdx.x, nthr = blockDim.x;
    int a;
    int64_t tile2;
    {
        // XCD x runs blocks x, x + 8, ... : 32 of them side by side.  They own 32 / A gene pairs x A chunks: a bucket's lists are
        // fetched from the fabric once per XCD and (pair, A chunks) share the Y chunks.  A = 1 (long lists: many permutations)
        // gives 32 pairs walking ONE chunk's lists; the split variant's short lists weigh less than the Y chunks and take A = 4.
        const uint32_t bid = blockIdx.x, x = bid & 7, k = bid >> 3, slot = k & 31, group = k >> 5;
        const uint32_t A = (uint32_t)a_per_xcd, pslots = 32u / A, nag = ((uint32_t)nch + A - 1u) / A;
        a = (int)((group % nag) * A + slot % A);
        tile2 = (int64_t)(group / nag) * (8 * pslots) + x * pslots + slot / A;
    }
    if (tile2 >= G2 || a >= nch) return;  // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg_raw = (int)blockIdx.y * (LDS_PERM_BLOCK / 64) + wave;
    const bool live = pg_raw < npg;  // waves past the last group only help with the chunk loads
    const int pg = live ? pg_raw : 0;
    const int64_t p = live ? (int64_t)pg * 64 + perm_slot(tid & 63) : pc;
    const double2* Zg = reinterpret_cast<const double2*>(Zp) + (size_t)tile2 * n;
    const double2* Yg = reinterpret_cast<const double2*>(Yp) + (size_t)tile2 * n;
    const bool staged = nthr * LDS_STAGE >= m;  // block-uniform: a whole chunk fits the threads' staging registers
    double2 sy[LDS_STAGE];
    double sr[LDS_STAGE];
#pragma unroll
    for (int u = 0; u < LDS_STAGE; ++u) {
        sy[u] = make_double2(0.0, 0.0);
        sr[u] = 0.0;
    }
    auto fetch = [&](int b) {  // chunk b of Y (and of the row sums) -> registers
        const int64_t j0 = (int64_t)b * m;
        const int mb = (int)min((int64_t)m, n - j0);
#pragma unroll
        for (int u = 0; u < LDS_STAGE; ++u) {
            const int t = tid + u * nthr;
            if (t < mb) {
                sy[u] = Yg[j0 + t];
                if (RARR) sr[u] = rowsum[j0 + t];
            }
        }
    };
    auto commit = [&](int b) {  // registers -> LDS
        const int mb = (int)min((int64_t)m, n - (int64_t)b * m);
#pragma unroll
        for (int u = 0; u < LDS_STAGE; ++u) {
            const int t = tid + u * nthr;
            if (t < mb) {
                Yc[t] = sy[u];
                if (RARR) Rc[t] = sr[u];
            }
        }
    };
    {
        const int64_t i0 = (int64_t)a * m;
        const int ma = (int)min((int64_t)m, n - i0);
        for (int t = tid; t < ma; t += nthr) Zc[t] = Zg[i0 + t];
        if (tid < ZERO_ROWS) Zc[m + tid] = make_double2(0.0, 0.0);
        if ((RCLS || REXC) && tid < 8) Rc[tid] = rowsum[tid];  // (`rowsum` points at the class table; RMODE 3: of r - r0)
    }
    if (staged) fetch(0);
    const size_t bk = ((size_t)pg * nch + a) * nch;
    const uint32_t* lst = lists + (size_t)base[pg] * 64 + (tid & 63);
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    for (int b = 0; b < nch; ++b) {
        __syncthreads();  // the previous Y chunk is no longer read (first pass: nothing to wait for but the Z stores)
        if (staged) {
            commit(b);
        } else {
            const int64_t j0 = (int64_t)b * m;
            const int mb = (int)min((int64_t)m, n - j0);
            for (int t = tid; t < mb; t += nthr) {
                Yc[t] = Yg[j0 + t];
                if (RARR) Rc[t] = rowsum[j0 + t];
            }
        }
        __syncthreads();
        if (staged && b + 1 < nch) fetch(b + 1);  // in flight while this chunk is consumed
        const uint32_t L = live ? (uint32_t)__builtin_amdgcn_readfirstlane((int)len[bk + b]) : 0u;
        const uint32_t* q = lst + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)off[bk + b]) * 64;
        uint32_t code[UNR];
        if (L > 0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) code[u] = q[u * 64];
        }
        for (uint32_t k = 0; k < L; k += UNR) {
            uint32_t cur[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) cur[u] = code[u];
            q += UNR * 64;
            if (k + UNR < L) {  // the next rows are in flight while these are consumed
#pragma unroll
                for (int u = 0; u < UNR; ++u) code[u] = q[u * 64];
            }
            double2 z[UNR], y[UNR];
            double r[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                z[u] = Zc[cur[u] & 0xffffu];
                y[u] = Yc[(cur[u] >> 16) & JMASK];
                if (RARR) r[u] = Rc[cur[u] >> 16];
                if (RCLS) r[u] = Rc[cur[u] >> 29];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                a0 = fma(z[u].x, y[u].x, a0);
                a1 = fma(z[u].y, y[u].y, a1);
                if (GEARY) {
                    b0 = fma(z[u].x * z[u].x, r[u], b0);
                    b1 = fma(z[u].y * z[u].y, r[u], b1);
                }
            }
        }
    }
    if (REXC && live) {  // the pairs whose j carries another row sum than most: z_i^2 (r[j] - r0), chunk a's Z rows are still in LDS
        const size_t xk = (size_t)pg * nch + a;
        const uint32_t Lx = (uint32_t)__builtin_amdgcn_readfirstlane((int)xlen[xk]);
        const uint32_t* q = xlists + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)xoff[xk]) * 64 + (tid & 63);
        for (uint32_t k = 0; k < Lx; k += EXC_UNROLL) {
            uint32_t code[EXC_UNROLL];
#pragma unroll
            for (int u = 0; u < EXC_UNROLL; ++u) code[u] = q[(size_t)(k + u) * 64];
#pragma unroll
            for (int u = 0; u < EXC_UNROLL; ++u) {
                const double2 z = Zc[code[u] & 0xffffu];
                const double r = Rc[code[u] >> 29];
                b0 = fma(z.x * z.x, r, b0);
                b1 = fma(z.y * z.y, r, b1);
            }
        }
    }
    if (p < pc) {
        const size_t o = (((size_t)tile2 * pc + p) * nch + a) * GP;
        *reinterpret_cast<double2*>(part1 + o) = make_double2(a0, a1);
        if (SECOND) *reinterpret_cast<double2*>(part2 + o) = make_double2(b0, b1);
    }
}

// S virtual permutations per permutation (LDS_SPLIT variant; 1 otherwise): their partial sums are added in the order (s, a).
// part2 == nullptr with GEARY: uniform row sums — the z^2 r term is rs_const * sum z^2 whatever the permutation.
template <bool GEARY>
__global__ __launch_bounds__(256) void k_perm_final_lds(const double* __restrict__ part1, const double* __restrict__ part2, int nch, int S,
                                                        int64_t pc, int64_t G, int64_t n, double W, const double* __restrict__ z2ss,
                                                        const double* __restrict__ qsum, const uint8_t* __restrict__ isconst,
                                                        double* __restrict__ sims, double rs_const) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    if (g >= G) return;
    const size_t o = (((size_t)(g >> 1) * pc * S + p * S) * nch) * GP + (g & 1);
    double s1 = 0.0, s2 = 0.0;
    for (int a = 0; a < nch * S; ++a) {  // (s, a) is contiguous: [vp = p * S + s][a]
        s1 += part1[o + (size_t)a * GP];
        if (GEARY && part2) s2 += part2[o + (size_t)a * GP];
    }
    if (GEARY && !part2) s2 = rs_const * z2ss[g];
    else if (GEARY && rs_const != 0.0) s2 += rs_const * z2ss[g];  // (the exception lists hold the departures from r0 = rs_const)
    double v;
    if (GEARY)
        v = ((double)(n - 1) * ((s2 - 2.0 * s1) + qsum[g])) / (2.0 * W * z2ss[g]);
    else
        v = (double)n / W * s1 / z2ss[g];
    sims[(size_t)p * G + g] = isconst[g] ? __builtin_nan("") : v;
}

// What gr/_ppatterns.py:474-492 takes out of the (P, G) permutation scores, formed where they are: per feature g
//   ge  = #{p : sims[p][g] >= score[g]}                                   `(sims >= score).sum(axis=0)`
//   sum = sims[:, g].sum()                                                `sims.sum(axis=0)`
//   var = mean(|x - sum / P|^2), std = sqrt(var)                          `np.var(sims, axis=0)`, `sims.std(axis=0)`
// bit for bit: numpy reduces the leading axis of a C-contiguous array by one rounded add per row, in row order
// (pairwise summation only applies along the contiguous axis), `_var` forms `arrmean = sum / P`, `x = arr - arrmean`,
// `x * x`, sums the same way and divides by P.  One thread per feature; every operation rounded separately.
//
// ONE feature in the whole call (`genes="x"`): the (P, 1) array is contiguous along the reduced axis too and numpy takes its
// other route — `pairwise_sum` of umath/loops_utils.h (fewer than 8 elements: in order; up to 128: eight interleaved partial
// sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail in order; longer: split at n/2 rounded down to a multiple
// of 8), applied to every buffer-sized run of 8192 elements, the runs added in order.  np_contiguous_sum restates that.
template <typename F>
__device__ double np_pairwise(int64_t lo, int64_t n, const F& at) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += at(lo + i);
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = at(lo + j);
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += at(lo + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += at(lo + i);
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(lo, n2, at) + np_pairwise(lo + n2, n - n2, at);
}
template <typename F>
__device__ double np_contiguous_sum(int64_t n, const F& at) {
    double res = 0.0;
    for (int64_t c = 0; c < n; c += 8192) res += np_pairwise(c, (n - c < 8192) ? n - c : 8192, at);
    return res;
}

__global__ __launch_bounds__(256) void k_perm_stats(const double* __restrict__ sims, int64_t P, int64_t G, const double* __restrict__ score,
                                                    long long* __restrict__ ge, double* __restrict__ sum, double* __restrict__ sd,
                                                    double* __restrict__ var, int contiguous_column) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double sc = score[g];
    if (contiguous_column) {  // G == 1 here
        long long cnt = 0;
        for (int64_t p = 0; p < P; ++p) cnt += (sims[p] >= sc) ? 1 : 0;
        const double total = np_contiguous_sum(P, [&](int64_t p) { return sims[p]; });
        const double m = total / (double)P;
        const double v = np_contiguous_sum(P, [&](int64_t p) { const double d = sims[p] - m; return d * d; }) / (double)P;
        ge[g] = cnt;
        sum[g] = total;
        var[g] = v;
        sd[g] = sqrt(v);
        return;
    }
    double acc = 0.0;
    long long cnt = 0;
#pragma unroll 8
    for (int64_t p = 0; p < P; ++p) {  // (the loads are independent of the add chain: eight in flight per thread)
        const double x = sims[(size_t)p * G + g];
        acc += x;
        cnt += (x >= sc) ? 1 : 0;
    }
    const double m = acc / (double)P;
    double acc2 = 0.0;
#pragma unroll 8
    for (int64_t p = 0; p < P; ++p) {
        const double d = sims[(size_t)p * G + g] - m;
        acc2 += d * d;
    }
    const double v = acc2 / (double)P;
    ge[g] = cnt;
    sum[g] = acc;
    var[g] = v;
    sd[g] = sqrt(v);
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

// The expression matrix resident on the device (uploaded once per call): dense row-major float64 / float32, or scipy's
// CSR / CSC arrays as they are (int64 indptr, int32 indices; values float32 or float64).
struct sqgr_matrix {
    sqgr_ctx* ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0, ld = 0;
    int kind = 0;        // 0 dense, 1 CSR (rows = cells), 2 CSC (columns = features)
    bool f32 = false;    // values are float32
    DevBuf<double> data;
    DevBuf<float> data32;
    DevBuf<int64_t> indptr;
    DevBuf<int32_t> indices;
    int64_t nnz = 0;
    // CSR matrices: the same entries by column, built on the device the first time a column LIST is asked for (ensure_by_column)
    int64_t cols_pending = -1;  // sqgr_matrix_alloc_dense: columns still to be uploaded (a streaming session is open while > 0)
    mutable bool by_col_ready = false;
    mutable DevBuf<int64_t> c_indptr;
    mutable DevBuf<int32_t> c_rows;
    mutable DevBuf<double> c_data;
    mutable DevBuf<float> c_data32;
    int ensure_by_column() const;
};

int sqgr_matrix::ensure_by_column() const {
    if (kind != 1 || by_col_ready) return SQGR_OK;
    hipStream_t st = ctx->stream;
    DevBuf<unsigned long long> cnt;
    SQGR_TRY(cnt.alloc((size_t)n_cols + 1));
    SQGR_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)n_cols + 1) * 8, st));
    const size_t count = (size_t)std::max<int64_t>(nnz, 1);
    SQGR_TRY(c_indptr.alloc((size_t)n_cols + 1));
    SQGR_TRY(c_rows.alloc_pooled(count));  // (every entry is written by the scatter)
    SQGR_TRY(f32 ? c_data32.alloc_pooled(count) : c_data.alloc_pooled(count));
    LaunchTimer t(ctx, "autocorr_csr_to_csc");
    if (nnz > 0) k_count_columns<<<(unsigned)ceil_div(nnz, 256), 256, 0, st>>>(indices.p, nnz, cnt.p);
    std::vector<unsigned long long> h((size_t)n_cols + 1);
    SQGR_HIP(hipMemcpyAsync(h.data(), cnt.p, ((size_t)n_cols + 1) * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> ptr((size_t)n_cols + 1);
    unsigned long long run = 0;
    for (int64_t c = 0; c <= n_cols; ++c) {  // exclusive scan; the cursors start at the column starts
        const unsigned long long v = c < n_cols ? h[c] : 0;
        ptr[c] = (int64_t)run;
        h[c] = run;
        run += v;
    }
    SQGR_HIP(hipMemcpyAsync(c_indptr.p, ptr.data(), ((size_t)n_cols + 1) * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(cnt.p, h.data(), ((size_t)n_cols + 1) * 8, hipMemcpyHostToDevice, st));
    if (nnz > 0) {
        if (f32) k_scatter_to_columns<float><<<(unsigned)ceil_div(n_rows, 4), 256, 0, st>>>(indptr.p, indices.p, data32.p, n_rows, cnt.p, c_rows.p, c_data32.p);
        else k_scatter_to_columns<double><<<(unsigned)ceil_div(n_rows, 4), 256, 0, st>>>(indptr.p, indices.p, data.p, n_rows, cnt.p, c_rows.p, c_data.p);
    }
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipStreamSynchronize(st));  // `cnt`, `h`, `ptr` are released on return
    by_col_ready = true;
    return SQGR_OK;
}

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
    DevBuf<double> sims_all;  // [P][G] scores of a whole permutation test kept on the device (sqgr_autocorr_perm_stats)
    PcgWorkspace pcg_ws;          // numpy-stream permutations generated in place (sqgr_autocorr_perms_pcg64)
    DevBuf<uint64_t> pcg_states;
    // LDS-bucketed permutation dot: gene-pair layout of Z and Y, and the bucket lists of the permutations in flight
    DevBuf<double> Zp, Yp;
    bool pairs_ready = false;
    // Geary's C under permutation needs sum_i z_i^2 r[idx_p(i)] with r = the graph's row sums.  On a row-normalised graph
    // (`transformation=True`, the reference's default, gr/_ppatterns.py:212-214) every row sum is 1 up to an ulp: the term is
    // r * sum z^2 for every permutation, and Geary's permutations run through the Moran kernel (two LDS reads per pair, not three)
    bool rs_uniform = false;
    double rs_const = 0.0;
    // ... and when they take a few distinct values (one per degree): class of every spot's row sum + the table (k_perm_dot_lds RMODE 2)
    int rs_classes = 0;  // 2..8: table mode available
    uint64_t cls_serial = 0;  // identifies this plan's class assignment (bucket lists carry the classes: part of their cache key)
    DevBuf<uint8_t> rcls;
    DevBuf<double> rtab;
    // ... and when one class holds most spots (a grid: all but its border): r0 * sum z^2 + exception lists (k_perm_dot_lds RMODE 3)
    bool rs_exc = false;
    int rs_main = 0;         // the class of r0
    double rs_main_val = 0.0;
    DevBuf<double> rtab_x;   // r_c - r0
    // what crosses PCIe per call — the observed scores in, G x 4 reductions out — goes through ONE persistent device block
    // [score | sum | std | var | n_ge] and its pinned twin, both made with the plan: no allocation, no pageable-copy staging and
    // one copy each way per sqgr_autocorr_perm_stats (round 5: three hipMalloc / hipFree and five pageable copies per call)
    DevBuf<double> stat_dev;
    double* stat_host = nullptr;
    int ensure_stat() {
        if (stat_dev.p && stat_host) return SQGR_OK;
        SQGR_TRY(stat_dev.alloc((size_t)5 * G));
        SQGR_HIP(hipHostMalloc(reinterpret_cast<void**>(&stat_host), (size_t)5 * G * sizeof(double), hipHostMallocDefault));
        return SQGR_OK;
    }
    ~sqgr_autocorr() {
        if (stat_host) (void)hipHostFree(stat_host);
    }
};

// The bucket lists depend on the permutations alone: every feature block of a call (and the next call with the same seed)
// walks the same ones, so they live in the context, keyed by what determines them.
struct PermLists : sqgr::CtxCache {
    // key
    int64_t n = -1, pc = 0, perm0 = 0;
    int m = 0, nch = 0, kind = -1;  // kind 0: device generator (seed), 1: numpy streams (states)
    int split = 1;                  // virtual permutations per permutation (LDS_SPLIT variant) | layout flags, see autocorr_perms
    uint64_t seed = 0;
    uint64_t cls_serial = 0;        // lists with row-sum classes in their entries: whose classes (sqgr_autocorr::cls_serial)
    std::vector<uint64_t> states;
    // lists
    DevBuf<uint32_t> b_len, b_off, b_total, lists, lists_raw;
    DevBuf<uint16_t> n_len;                  // [group][a][b][lane] pairs in the lane's list (the schedule kernels choose by them)
    DevBuf<uint16_t> stp_scr;                // scratch of k_bucket_order_steps, shaped like the lists: the step given to every pair
    DevBuf<uint32_t> x_len, x_off, x_lists;  // exception lists (see k_exc_offsets), when the lists serve RMODE 3
    DevBuf<uint64_t> b_base;
    bool matches(int64_t n_, int64_t pc_, int64_t perm0_, int m_, int nch_, int kind_, uint64_t seed_, const uint64_t* st_, int split_,
                 uint64_t cls_serial_) const {
        if (n != n_ || pc != pc_ || perm0 != perm0_ || m != m_ || nch != nch_ || kind != kind_ || split != split_ || cls_serial != cls_serial_)
            return false;
        if (kind == 0) return seed == seed_;
        return kind == 1 && st_ && states.size() == (size_t)pc_ * 4 && !memcmp(states.data(), st_, (size_t)pc_ * 32);
    }
};

static PermLists* perm_lists(sqgr_ctx* ctx) {
    if (!ctx->autocorr_lists) ctx->autocorr_lists = new PermLists();
    return static_cast<PermLists*>(ctx->autocorr_lists);
}

// 0: the gather kernel (k_perm_dot), 1: the LDS-bucketed kernel.  SQGR_AUTOCORR_KERNEL=gather|lds overrides the choice
// (tests run both); the LDS kernel needs <= LDS_MAX_CHUNKS chunks and pays off with many permutations per gene block.
// 2: the LDS kernel on LDS_SPLIT virtual permutations per permutation (fewer than 512 permutations; SQGR_AUTOCORR_KERNEL=lds-split).
static int perm_kernel_choice(int64_t n, int64_t G, int64_t P, bool geary) {
    int nch = 0;
    (void)lds_chunk(n, geary, &nch);
    if (nch > LDS_MAX_CHUNKS) return 0;
    if (const char* env = getenv("SQGR_AUTOCORR_KERNEL")) {
        if (!strcmp(env, "lds")) return 1;
        if (!strcmp(env, "lds-split")) return 2;
        if (!strcmp(env, "gather")) return 0;
    }
    if (G < 256 || n < 4096) return 0;
    // (measured at config 3's shape, 2048 genes: the Y chunks every workgroup streams through LDS cost the same however few
    // permutations there are — the gather kernel wins below ~40 permutations)
    return P >= 512 ? 1 : (P >= 40 ? 2 : 0);
}

// How the bucket lists are ordered.  0: as built (ascending i; SQGR_AUTOCORR_ORDER_LISTS=0, and the split variant: a sub-list holds
// two pairs per class), 1: SQGR_AUTOCORR_ORDER=single, the round-3 schedule (every list on its own: one side of a pair conflict-free),
// 2: SQGR_AUTOCORR_ORDER=rotation, the joint schedule of the 16 permutations of a `ds_read_b128` lane group on the fixed rotation of
// Z classes (k_bucket_order_joint), 3: the step schedule (k_bucket_order_steps; the default)
static int list_order_mode(int split) {
    if (const char* e = getenv("SQGR_AUTOCORR_ORDER_LISTS"))
        if (atoi(e) == 0) return 0;
    const char* e_order = getenv("SQGR_AUTOCORR_ORDER");
    // (the split variant — lane = (permutation, spots i = s mod 8), sub-lists of ~30 pairs on two Z classes each — takes the joint
    // schedules like any other list: a lane group is two permutations x 8 sub-lists; round 3's per-list order has no use for it)
    if (e_order && !strcmp(e_order, "single")) return split == 1 ? 1 : 0;
    if (e_order && !strcmp(e_order, "rotation")) return 2;
    return 3;
}

// bucket lists of the pc permutations whose indices are in idx (the first one is permutation `perm0` of its stream) -> pl
// (split > 1: pc counts VIRTUAL permutations, idx holds pc / split index rows)
static int build_perm_lists(sqgr_ctx* ctx, PermLists* pl, const int32_t* idx, int64_t n, int64_t pc, int64_t perm0, int m, int nch, int split,
                            bool geary, const uint8_t* rcls, const uint8_t* exc_cls = nullptr, int exc_main = 0) {
    hipStream_t st = ctx->stream;
    const int npg = (int)ceil_div(pc, 64);
    const int order = list_order_mode(split);
    const bool order_lists = order != 0;
    const bool joint = order >= 2 && (perm0 * split) % 16 == 0;  // (the lane groups must be those of the global permutation index)
    const int round = (split > 1 && !joint) ? LIST_ROUND_SPLIT : LIST_ROUND;
    const int nb = nch * nch;
    pl->n = -1;  // invalid until complete
    SQGR_TRY(pl->b_len.ensure((size_t)npg * nb));
    SQGR_TRY(pl->b_off.ensure((size_t)npg * nb));
    SQGR_TRY(pl->b_total.ensure((size_t)npg * 2));
    SQGR_TRY(pl->b_base.ensure((size_t)npg + 2));
    const size_t cnt_lds = (size_t)nch * 64 * sizeof(uint32_t);
    uint64_t rows_max[2] = {0, 0};  // all rows, the longest list
    LaunchTimer t(ctx, "autocorr_bucket_lists");
    const int xcnt = npg * nch;
    uint32_t xrows = 0;
    if (exc_cls) {  // the pairs whose j has another row sum than most (RMODE 3) get lists of their own, built in the same two passes
        SQGR_TRY(pl->x_len.ensure((size_t)xcnt));
        SQGR_TRY(pl->x_off.ensure((size_t)xcnt + 1));
    }
    k_bucket_count<<<dim3((unsigned)nch, (unsigned)npg), 64, cnt_lds, st>>>(idx, n, pc, m, nch, split, round, pl->b_len.p, exc_cls, exc_main, pl->x_len.p);
    k_bucket_offsets<<<(unsigned)npg, 256, 0, st>>>(pl->b_len.p, nb, pl->b_off.p, pl->b_total.p);
    k_bucket_bases<<<1, 64, 0, st>>>(pl->b_total.p, npg, pl->b_base.p);
    if (exc_cls) k_exc_offsets<<<1, 256, 0, st>>>(pl->x_len.p, xcnt, pl->x_off.p);
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipMemcpyAsync(rows_max, pl->b_base.p + npg, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (exc_cls) SQGR_HIP(hipMemcpyAsync(&xrows, pl->x_off.p + xcnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    if (exc_cls && (pl->x_lists.n < (size_t)xrows * 64 + 64 || !pl->x_lists.p)) SQGR_TRY(pl->x_lists.alloc_pooled((size_t)xrows * 64 + (size_t)xrows * 4 + 64));
    const uint64_t rows = rows_max[0];
    // (with headroom: the next seed's lists are a few rows longer or shorter, and a new buffer of this size costs milliseconds)
    if (pl->lists.n < (size_t)rows * 64 || !pl->lists.p) SQGR_TRY(pl->lists.alloc_pooled((size_t)rows * 64 + (size_t)rows * 4));
    uint32_t* fill_to = pl->lists.p;
    if (joint) {  // out of place: the lists as built -> lists_raw, scheduled -> lists
        SQGR_TRY(pl->n_len.ensure((size_t)npg * nb * 64));
        if (pl->lists_raw.n < (size_t)rows * 64 || !pl->lists_raw.p) SQGR_TRY(pl->lists_raw.alloc_pooled((size_t)rows * 64 + (size_t)rows * 4));
        if (order == 3 && (pl->stp_scr.n < (size_t)rows * 64 || !pl->stp_scr.p)) SQGR_TRY(pl->stp_scr.alloc_pooled((size_t)rows * 64 + (size_t)rows * 4));
        fill_to = pl->lists_raw.p;
    }
    k_bucket_fill<<<dim3((unsigned)nch, (unsigned)npg), 64, cnt_lds + (size_t)nch * sizeof(uint32_t), st>>>(idx, n, pc, m, nch, split, pl->b_len.p, pl->b_off.p,
                                                                                                          pl->b_base.p, fill_to, rcls, exc_cls, exc_main,
                                                                                                          pl->x_len.p, pl->x_off.p, pl->x_lists.p,
                                                                                                          joint ? pl->n_len.p : nullptr);
    SQGR_HIP(hipGetLastError());
    if (joint) {
        // the step schedule takes the lane groups whose longest list fits STEP_SEG rows, the rotation schedule the others (in segments)
        if (order == 3 && rows_max[1] > 0)
            k_bucket_order_steps<<<dim3((unsigned)nb, (unsigned)npg), 64, 0, st>>>(m, nb, pl->b_len.p, pl->b_off.p, pl->b_base.p, pl->lists_raw.p,
                                                                                 pl->lists.p, pl->n_len.p, pl->stp_scr.p);
        const int longer_than = order == 3 ? STEP_SEG : 0;
        const int segs = (int64_t)rows_max[1] > longer_than ? (int)ceil_div((int64_t)rows_max[1], (int64_t)ORDER_SEG) : 0;
        if (segs > 0)
            k_bucket_order_joint<<<dim3((unsigned)nb * 4u, (unsigned)npg, (unsigned)segs), 16, 0, st>>>(m, nb, pl->b_len.p, pl->b_off.p, pl->b_base.p,
                                                                                                      pl->lists_raw.p, pl->lists.p, pl->n_len.p, longer_than);
        SQGR_HIP(hipGetLastError());
    } else if (order_lists && split == 1) {  // (the schedule works in rounds of 16 row classes; a sub-list of the split variant holds 2)
        // Geary's C reads TWO arrays through the Y index (the y row and the row sum): its lists are scheduled by the classes of the Y
        // row (measured 99.6 -> 94.4 ms per 2048 genes x 1000 permutations); Moran's I keeps the Z classes (66.2 vs 65.6 ms: no difference)
        const char* e_by = getenv("SQGR_AUTOCORR_ORDER_BY");
        const int cls_shift = e_by ? ((e_by[0] == 'y') ? 16 : 0) : (geary ? 16 : 0);
        k_bucket_order<<<dim3((unsigned)nb, (unsigned)npg), 64, 0, st>>>(m, nb, perm0, pl->b_len.p, pl->b_off.p, pl->b_base.p, pl->lists.p, cls_shift);
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
}

// which k_perm_dot_lds instantiation serves a statistic on this plan: 0 Moran's I — and Geary's C when all row sums are equal —,
// 3 Geary's C as Moran's loop + exception lists (<= 8 distinct row sums, one of them on at least 3 spots in 4), 2 Geary's C through
// the class table (<= 8 distinct row sums), 1 Geary's C with the row sums of the chunk in LDS
static int lds_rmode(const sqgr_autocorr* h, int32_t mode) {
    if (mode != 1 || h->rs_uniform) return 0;
    return h->rs_classes ? (h->rs_exc ? 3 : 2) : 1;
}

// permutation scores of the pc permutations behind the bucket lists `pl`, through the LDS-bucketed kernel -> h->sims
static int perms_pass_lds(sqgr_autocorr* h, int32_t mode, int64_t pc_real, const PermLists* pl, int split) {
    sqgr_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    const int64_t n = h->n, G = h->G, G2 = (G + 1) / 2;
    const bool geary_stat = mode == 1;
    const int rmode = lds_rmode(h, mode);
    const bool geary = rmode == 1;   // the kernel with the chunk's row sums in LDS: 4092-spot chunks
    const bool second = rmode != 0;  // a second partial sum (z^2 r) per lane
    int nch = 0;
    const int m = lds_chunk(n, geary, &nch);
    const int64_t pc = pc_real * split;  // lanes of the dot kernel: virtual permutations
    const int npg = (int)ceil_div(pc, 64);
    if (!h->pairs_ready) {
        SQGR_TRY(h->Zp.ensure((size_t)G2 * n * GP));
        SQGR_TRY(h->Yp.ensure((size_t)G2 * n * GP));
        LaunchTimer t(ctx, "autocorr_repack");
        const int64_t total = h->ntiles * n * GT;
        k_repack_pairs<<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(h->Zt.p, n, h->ntiles, G2, h->Zp.p);
        k_repack_pairs<<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(h->Yt.p, n, h->ntiles, G2, h->Yp.p);
        SQGR_HIP(hipGetLastError());
        h->pairs_ready = true;
    }
    SQGR_TRY(h->part1.ensure((size_t)G2 * pc * nch * GP));
    if (second) SQGR_TRY(h->part2.ensure((size_t)G2 * pc * nch * GP));
    const size_t lds = ((size_t)(2 * m + ZERO_ROWS) * GP + (geary ? (size_t)m : (rmode >= 2 ? (size_t)8 : 0))) * sizeof(double);
    // the split variant always launches whole workgroups: waves without a permutation group still move the Y chunks (with fewer
    // than m / 5 threads a chunk does not fit the staging registers and its loads are no longer prefetched)
    const int threads = split > 1 ? LDS_PERM_BLOCK : 64 * std::min(npg, LDS_PERM_BLOCK / 64);
    // chunks per XCD round (see k_perm_dot_lds): the power of two nearest sqrt(32 * Y chunk bytes / list bytes of one bucket)
    int a_per_xcd = 1;
    if (const char* env = getenv("SQGR_AUTOCORR_XCD_CHUNKS")) {
        a_per_xcd = std::max(1, std::min(32, atoi(env)));
        while (a_per_xcd & (a_per_xcd - 1)) a_per_xcd &= a_per_xcd - 1;
    } else if (split == 1) {
        // many permutations (measured at config 3's shape, 1000 permutations): Moran's I 66.2 ms with one chunk per round, 63.3 with 2,
        // 62.4 with 4; Geary's C (24 instead of 16 LDS bytes per pair, 4092-spot chunks) 99.6 / 100.6 / 106.9 ms
        a_per_xcd = geary ? 1 : 4;
    } else {
        const double list_bytes = (double)npg * ((double)n / split / ((double)nch * nch) * 1.5) * 256.0;  // rows of 64 entries, ~+50 % padding
        // (measured at config 3's shape, 50 / 100 / 256 permutations: 4-8 chunks per round, -14 ... -27 % against one; twice the
        // square-root estimate, at most 8)
        const double want = 2.0 * std::sqrt(32.0 * (double)m * GP * 8 / std::max(list_bytes, 1.0));
        while (a_per_xcd * 2 <= 8 && (double)a_per_xcd < want) a_per_xcd *= 2;
    }
    const int pslots = 32 / a_per_xcd;
    dim3 grid((unsigned)(ceil_div(G2, 8 * pslots) * ceil_div(nch, a_per_xcd) * 256), (unsigned)ceil_div(npg, LDS_PERM_BLOCK / 64));
    {
        LaunchTimer t(ctx, split > 1 ? (geary_stat ? "autocorr_perm_dot_lds_split_geary" : "autocorr_perm_dot_lds_split_moran")
                                     : (geary_stat ? "autocorr_perm_dot_lds_geary" : "autocorr_perm_dot_lds_moran"));
#define SQGR_DOT_LDS(RM, RSRC, P2)                                                                                                            \
    do {                                                                                                                                      \
        if (lds > 64 * 1024)                                                                                                                  \
            SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_perm_dot_lds<RM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_perm_dot_lds<RM><<<grid, threads, lds, st>>>(h->Zp.p, h->Yp.p, RSRC, n, pc, npg, G2, m, nch, pl->b_len.p, pl->b_off.p, pl->b_base.p,  \
                                                       pl->lists.p, h->part1.p, P2, a_per_xcd, pl->x_len.p, pl->x_off.p, pl->x_lists.p);     \
    } while (0)
        if (rmode == 1) SQGR_DOT_LDS(1, h->rowsum.p, h->part2.p);
        else if (rmode == 2) SQGR_DOT_LDS(2, h->rtab.p, h->part2.p);
        else if (rmode == 3) SQGR_DOT_LDS(3, h->rtab_x.p, h->part2.p);
        else SQGR_DOT_LDS(0, h->rowsum.p, nullptr);
#undef SQGR_DOT_LDS
        SQGR_HIP(hipGetLastError());
    }
    {
        LaunchTimer t(ctx, "autocorr_perm_final");
        dim3 g2((unsigned)ceil_div(G, 256), (unsigned)pc_real);
        if (geary_stat)
            k_perm_final_lds<true><<<g2, 256, 0, st>>>(h->part1.p, second ? h->part2.p : nullptr, nch, split, pc_real, G, n, h->W, h->z2ss.p, h->qsum.p,
                                                       h->isconst.p, h->sims.p, rmode == 3 ? h->rs_main_val : (second ? 0.0 : h->rs_const));
        else
            k_perm_final_lds<false><<<g2, 256, 0, st>>>(h->part1.p, nullptr, nch, split, pc_real, G, n, h->W, h->z2ss.p, h->qsum.p, h->isconst.p, h->sims.p,
                                                        0.0);
        SQGR_HIP(hipGetLastError());
    }
    return SQGR_OK;
}

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
                           const sqgr_matrix* dm = nullptr, int64_t dev_col0 = 0, const int32_t* dev_cols = nullptr) {
    // dev_cols != NULL: the features are columns dev_cols[0 .. G) of the resident matrix (a device array), else [dev_col0, dev_col0 + G)
    const double* dev_x = (dm && dm->kind == 0 && !dm->f32 && !dev_cols) ? dm->data.p : nullptr;
    const int64_t dev_ld = dm ? dm->ld : 0;
    SQGR_REQUIRE(ctx && g && (vals || dm) && out, "null argument");
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
    // stage the gene-major input in blocks of <= 256 genes (and <= ~256 MB) when it comes from the host; a block cut out of a matrix
    // that is resident on the device is staged whole (<= 2 GB: no wait for the staging buffer between its pieces — seven stream
    // synchronisations per 2048-gene block of config 3)
    int64_t gc_max = std::max<int64_t>(GT, std::min<int64_t>(256, ((int64_t)1 << 28) / (n * 8) / GT * GT));
    if (dm && (int64_t)n * G * 8 <= ((int64_t)2 << 30)) gc_max = std::max<int64_t>(gc_max, ceil_div(G, GT) * GT);
    DevBuf<double> X, D;
    if ((rc = X.alloc_pooled((size_t)gc_max * n))) return fail(rc);
    hipError_t e = hipSuccess;
    const double* cm_src = dev_x ? dev_x + dev_col0 : nullptr;  // cell-major source on the device and its row pitch
    int64_t cm_ld = dev_x ? dev_ld : G;
    if (cell_major && !dm) {  // one contiguous upload of vals[n][G]; the gene blocks are cut out of it on the device
        if ((rc = D.alloc((size_t)n * G))) return fail(rc);
        e = hipMemcpyAsync(D.p, vals, (size_t)n * G * 8, hipMemcpyHostToDevice, st);
        cm_src = D.p;
    }
    for (int64_t g0 = 0; g0 < G && e == hipSuccess; g0 += gc_max) {
        const int gc = (int)std::min<int64_t>(gc_max, G - g0);
        if (dm && dev_cols) {  // a LIST of columns of the resident matrix
            LaunchTimer t(ctx, "autocorr_expand");
            const dim3 tgrid((unsigned)ceil_div(n, GT), (unsigned)ceil_div(gc, GT));
            if (dm->kind == 0) {
                if (dm->f32) k_cells_to_genes_idx<float><<<tgrid, 256, 0, st>>>(dm->data32.p, dm->ld, n, dev_cols + g0, gc, X.p);
                else k_cells_to_genes_idx<double><<<tgrid, 256, 0, st>>>(dm->data.p, dm->ld, n, dev_cols + g0, gc, X.p);
            } else {
                e = hipMemsetAsync(X.p, 0, (size_t)gc * n * 8, st);
                if (e != hipSuccess) break;
                const bool twin = dm->kind == 1;  // CSR: its by-column twin (built by the caller of this function)
                const int64_t* cp = twin ? dm->c_indptr.p : dm->indptr.p;
                const int32_t* cr = twin ? dm->c_rows.p : dm->indices.p;
                const dim3 cgrid(8, (unsigned)gc);
                if (dm->f32) k_csc_to_genes_idx<float><<<cgrid, 256, 0, st>>>(cp, cr, twin ? dm->c_data32.p : dm->data32.p, n, dev_cols + g0, X.p);
                else k_csc_to_genes_idx<double><<<cgrid, 256, 0, st>>>(cp, cr, twin ? dm->c_data.p : dm->data.p, n, dev_cols + g0, X.p);
            }
            e = hipGetLastError();
        } else if (dm && !dev_x) {  // float32 and / or sparse resident matrix: the gene block is formed from it on the device
            LaunchTimer t(ctx, "autocorr_expand");
            const dim3 tgrid((unsigned)ceil_div(n, GT), (unsigned)ceil_div(gc, GT));
            if (dm->kind == 0) {
                k_cells_to_genes_f32<<<tgrid, 256, 0, st>>>(dm->data32.p + dev_col0, dm->ld, n, g0, gc, X.p);
            } else {
                e = hipMemsetAsync(X.p, 0, (size_t)gc * n * 8, st);
                if (e != hipSuccess) break;
                if (dm->kind == 1) {
                    if (dm->f32) k_csr_to_genes<float><<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(dm->indptr.p, dm->indices.p, dm->data32.p, n, dev_col0 + g0, gc, X.p);
                    else k_csr_to_genes<double><<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(dm->indptr.p, dm->indices.p, dm->data.p, n, dev_col0 + g0, gc, X.p);
                } else {
                    const dim3 cgrid(8, (unsigned)gc);
                    if (dm->f32) k_csc_to_genes<float><<<cgrid, 256, 0, st>>>(dm->indptr.p, dm->indices.p, dm->data32.p, n, dev_col0 + g0, X.p);
                    else k_csc_to_genes<double><<<cgrid, 256, 0, st>>>(dm->indptr.p, dm->indices.p, dm->data.p, n, dev_col0 + g0, X.p);
                }
            }
            e = hipGetLastError();
        } else if (cm_src) {
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
    {   // distinct values of the row sums (exact comparison): 1 -> the z^2 r term of Geary's permutations is a constant; <= 8 -> classes
        const char* env = getenv("SQGR_AUTOCORR_ROWSUM_CLASSES");
        const bool allow = !(env && atoi(env) == 0);
        double vals8[8];
        int nv = 0;
        std::vector<uint8_t> cls((size_t)std::max<int64_t>(n, 1), 0);
        for (int64_t i = 0; allow && i < n && nv <= 8; ++i) {
            int c = 0;
            while (c < nv && vals8[c] != rs[i]) ++c;
            if (c == nv) {
                if (nv == 8) {
                    nv = 9;
                    break;
                }
                vals8[nv++] = rs[i];
            }
            cls[i] = (uint8_t)c;
        }
        h->rs_uniform = allow && n > 0 && nv == 1;
        h->rs_const = nv >= 1 ? vals8[0] : 0.0;
        h->rs_classes = (allow && nv >= 2 && nv <= 8) ? nv : 0;
        if (h->rs_classes) {
            static std::atomic<uint64_t> serial{0};
            h->cls_serial = ++serial;
            double tab[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tab_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = 0; c < nv; ++c) tab[c] = vals8[c];
            // one class on at least 3 spots in 4 (SQGR_AUTOCORR_ROWSUM_EXCEPTIONS=<max share of the others, default 0.25>; 0 disables):
            // its row sum becomes a constant term, the other spots' departures from it go through exception lists
            int64_t pop[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int64_t i = 0; i < n; ++i) pop[cls[i]] += 1;
            int cmain = 0;
            for (int c = 1; c < nv; ++c)
                if (pop[c] > pop[cmain]) cmain = c;
            double max_share = 0.25;
            if (const char* ex = getenv("SQGR_AUTOCORR_ROWSUM_EXCEPTIONS")) max_share = atof(ex);
            h->rs_exc = max_share > 0.0 && (double)(n - pop[cmain]) <= max_share * (double)n;
            h->rs_main = cmain;
            h->rs_main_val = vals8[cmain];
            for (int c = 0; c < nv; ++c) tab_x[c] = vals8[c] - vals8[cmain];
            if ((rc = h->rcls.alloc((size_t)n)) || (rc = h->rtab.alloc(8)) || (rc = h->rtab_x.alloc(8))) return fail(rc);
            e = hipMemcpyAsync(h->rcls.p, cls.data(), (size_t)n, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(h->rtab.p, tab, sizeof(tab), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(h->rtab_x.p, tab_x, sizeof(tab_x), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                set_error("autocorr prepare failed: %s", hipGetErrorString(e));
                return fail(SQGR_ERR_HIP);
            }
        }
    }
    if ((rc = h->ensure_stat())) return fail(rc);
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

int sqgr_matrix_create_dense(sqgr_ctx* ctx, const void* x, int32_t value_bytes, int64_t n_rows, int64_t n_cols, int64_t ld,
                             sqgr_matrix** out) {
    SQGR_REQUIRE(ctx && x && out && n_rows > 0 && n_cols > 0, "null argument or empty matrix");
    SQGR_REQUIRE(value_bytes == 4 || value_bytes == 8, "value_bytes must be 4 (float32) or 8 (float64), found %d", value_bytes);
    SQGR_REQUIRE(ld >= n_cols, "row pitch %lld < %lld columns", (long long)ld, (long long)n_cols);
    *out = nullptr;
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_matrix* m = new sqgr_matrix();
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->ld = n_cols;  // stored densely whatever the pitch of the source
    m->f32 = value_bytes == 4;
    const size_t count = (size_t)n_rows * n_cols;
    int rc = m->f32 ? m->data32.alloc_pooled(count) : m->data.alloc_pooled(count);  // (overwritten whole by the upload)
    if (rc != SQGR_OK) {
        delete m;
        return rc;
    }
    void* dst = m->f32 ? (void*)m->data32.p : (void*)m->data.p;
    hipError_t e = (ld == n_cols) ? hipMemcpyAsync(dst, x, count * value_bytes, hipMemcpyHostToDevice, ctx->stream)
                                  : hipMemcpy2DAsync(dst, (size_t)n_cols * value_bytes, x, (size_t)ld * value_bytes, (size_t)n_cols * value_bytes,
                                                     (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("matrix upload failed: %s", hipGetErrorString(e));
        delete m;
        return SQGR_ERR_HIP;
    }
    *out = m;
    return SQGR_OK;
}

// The same matrix filled column block by column block WHILE the first blocks are being worked on: sqgr_matrix_alloc_dense reserves
// the device array, sqgr_matrix_upload_columns copies a column range of the host matrix into it on the context's copy stream and
// waits for that stream only — it may be called from another host thread than the one that runs the statistics (the pool and the
// error state are thread-safe; nothing else of the context is touched).  16 GB of dense float64 cross PCIe in 0.29 s; uploaded
// whole before the first feature block, config 3 took 0.29 + 0.58 s.
int sqgr_matrix_alloc_dense(sqgr_ctx* ctx, int32_t value_bytes, int64_t n_rows, int64_t n_cols, sqgr_matrix** out) {
    SQGR_REQUIRE(ctx && out && n_rows > 0 && n_cols > 0, "null argument or empty matrix");
    SQGR_REQUIRE(value_bytes == 4 || value_bytes == 8, "value_bytes must be 4 (float32) or 8 (float64), found %d", value_bytes);
    *out = nullptr;
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_matrix* m = new sqgr_matrix();
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->ld = n_cols;
    m->f32 = value_bytes == 4;
    const size_t count = (size_t)n_rows * n_cols;
    const int rc = m->f32 ? m->data32.alloc_pooled(count) : m->data.alloc_pooled(count);  // (every column is uploaded before it is read)
    if (rc != SQGR_OK) {
        delete m;
        return rc;
    }
    m->cols_pending = n_cols;
    streaming_upload_begin();  // closed by the upload of the last column (or by sqgr_matrix_destroy)
    *out = m;
    return SQGR_OK;
}

int sqgr_matrix_upload_columns(sqgr_matrix* m, const void* x, int64_t ld, int64_t col0, int64_t n_cols) {
    SQGR_REQUIRE(m && x && m->kind == 0, "null argument or not a dense matrix");
    SQGR_REQUIRE(col0 >= 0 && n_cols > 0 && col0 + n_cols <= m->n_cols && ld >= n_cols, "columns [%lld, %lld) of %lld, row pitch %lld", (long long)col0,
                 (long long)(col0 + n_cols), (long long)m->n_cols, (long long)ld);
    sqgr_ctx* ctx = m->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    const size_t vb = m->f32 ? 4 : 8;
    char* dst = (m->f32 ? reinterpret_cast<char*>(m->data32.p) : reinterpret_cast<char*>(m->data.p)) + (size_t)col0 * vb;
    hipError_t e = hipMemcpy2DAsync(dst, (size_t)m->n_cols * vb, x, (size_t)ld * vb, (size_t)n_cols * vb, (size_t)m->n_rows, hipMemcpyHostToDevice,
                                    ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
    if (e != hipSuccess) {
        set_error("matrix upload failed: %s", hipGetErrorString(e));
        return SQGR_ERR_HIP;
    }
    if (m->cols_pending > 0) {
        m->cols_pending -= std::min<int64_t>(n_cols, m->cols_pending);
        if (m->cols_pending == 0) streaming_upload_end();
    }
    return SQGR_OK;
}

int sqgr_matrix_create(sqgr_ctx* ctx, const double* x, int64_t n_rows, int64_t n_cols, sqgr_matrix** out) {
    return sqgr_matrix_create_dense(ctx, x, 8, n_rows, n_cols, n_cols, out);
}

// kind 1: CSR (indptr has n_rows + 1 entries, indices are columns), kind 2: CSC (n_cols + 1, rows)
static int matrix_create_sparse(sqgr_ctx* ctx, int kind, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* indptr, const void* indices,
                                int32_t index_bytes, const void* values, int32_t value_bytes, sqgr_matrix** out) {
    SQGR_REQUIRE(ctx && indptr && out && n_rows > 0 && n_cols > 0 && nnz >= 0, "null argument or empty matrix");
    SQGR_REQUIRE((indices && values) || nnz == 0, "indices/values is NULL");
    SQGR_REQUIRE(index_bytes == 4 || index_bytes == 8, "index_bytes must be 4 or 8, found %d", index_bytes);
    SQGR_REQUIRE(value_bytes == 4 || value_bytes == 8, "value_bytes must be 4 (float32) or 8 (float64), found %d", value_bytes);
    SQGR_REQUIRE(n_rows < (int64_t)0x7fffffff && n_cols < (int64_t)0x7fffffff, "more than 2^31 rows or columns");
    *out = nullptr;
    const int64_t nptr = (kind == 1 ? n_rows : n_cols) + 1, minor = kind == 1 ? n_cols : n_rows;
    // host-side validation (the arrays are caller memory): monotone pointers ending at nnz, indices inside the minor axis,
    // ascending inside every major slice (the CSR expansion bisects its rows)
    auto at = [&](const void* a, int64_t i) -> int64_t {
        return index_bytes == 4 ? (int64_t) static_cast<const int32_t*>(a)[i] : static_cast<const int64_t*>(a)[i];
    };
    SQGR_REQUIRE(at(indptr, 0) == 0 && at(indptr, nptr - 1) == nnz, "indptr must run from 0 to nnz=%lld", (long long)nnz);
    for (int64_t j = 0; j + 1 < nptr; ++j)
        SQGR_REQUIRE(at(indptr, j) <= at(indptr, j + 1) && at(indptr, j + 1) <= nnz, "indptr is not monotone at %lld", (long long)j);
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    sqgr_matrix* m = new sqgr_matrix();
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->ld = n_cols;
    m->kind = kind;
    m->f32 = value_bytes == 4;
    m->nnz = nnz;
    auto fail = [&](int code) {
        delete m;
        return code;
    };
    int rc = SQGR_OK;
    const size_t cnt = (size_t)std::max<int64_t>(nnz, 1);
    if ((rc = m->indptr.alloc((size_t)nptr)) || (rc = m->indices.alloc_pooled(cnt)) || (rc = m->f32 ? m->data32.alloc_pooled(cnt) : m->data.alloc_pooled(cnt))) return fail(rc);
    hipError_t e = hipSuccess;
    if (index_bytes == 8) {
        DevBuf<int64_t> wide;
        if ((rc = wide.alloc(cnt))) return fail(rc);
        e = hipMemcpyAsync(m->indptr.p, indptr, (size_t)nptr * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(wide.p, indices, (size_t)nnz * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && nnz > 0) k_narrow_indices<<<(unsigned)ceil_div(nnz, 256), 256, 0, st>>>(wide.p, nnz, m->indices.p);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // `wide` is released on return
    } else {
        DevBuf<int32_t> narrow;
        if ((rc = narrow.alloc((size_t)nptr))) return fail(rc);
        e = hipMemcpyAsync(narrow.p, indptr, (size_t)nptr * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) k_widen_indptr<<<(unsigned)ceil_div(nptr, 256), 256, 0, st>>>(narrow.p, nptr, m->indptr.p);
        if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(m->indices.p, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess && nnz > 0)
        e = hipMemcpyAsync(m->f32 ? (void*)m->data32.p : (void*)m->data.p, values, (size_t)nnz * value_bytes, hipMemcpyHostToDevice, st);
    // the indices are checked where they now are: inside the minor axis, ascending inside every slice
    DevBuf<int> flags;
    int hflags[3] = {0, 0, 0};
    if (e == hipSuccess && (rc = flags.alloc(3))) return fail(rc);
    if (e == hipSuccess) e = hipMemsetAsync(flags.p, 0, 12, st);
    if (e == hipSuccess && nnz > 0) k_check_sparse<<<(unsigned)ceil_div(nptr - 1, 4), 256, 0, st>>>(m->indptr.p, m->indices.p, nptr - 1, minor, flags.p);
    if (e == hipSuccess) e = hipMemcpyAsync(hflags, flags.p, 12, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("sparse matrix upload failed: %s", hipGetErrorString(e));
        return fail(SQGR_ERR_HIP);
    }
    if (hflags[0] || hflags[1] || hflags[2]) {
        // (repeated indices: scipy's `toarray()` sums them in the matrix dtype, which a float64 accumulation would not reproduce
        //  for float32 values — the caller sums them first)
        set_error(hflags[0] ? "sparse matrix: an index lies outside [0,%lld)"
                            : (hflags[1] ? "sparse matrix: indices are not sorted inside a row/column (call .sort_indices())"
                                         : "sparse matrix: duplicate entries (call .sum_duplicates())"),
                  (long long)minor);
        return fail(SQGR_ERR_INVALID);
    }
    *out = m;
    return SQGR_OK;
}

int sqgr_matrix_create_csr(sqgr_ctx* ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* indptr, const void* indices,
                           int32_t index_bytes, const void* values, int32_t value_bytes, sqgr_matrix** out) {
    return matrix_create_sparse(ctx, 1, n_rows, n_cols, nnz, indptr, indices, index_bytes, values, value_bytes, out);
}

int sqgr_matrix_create_csc(sqgr_ctx* ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* indptr, const void* indices,
                           int32_t index_bytes, const void* values, int32_t value_bytes, sqgr_matrix** out) {
    return matrix_create_sparse(ctx, 2, n_rows, n_cols, nnz, indptr, indices, index_bytes, values, value_bytes, out);
}

int sqgr_matrix_destroy(sqgr_matrix* m) {
    if (!m) return SQGR_OK;
    (void)hipSetDevice(m->ctx->device);
    if (m->cols_pending > 0) streaming_upload_end();  // a streaming session that was never completed
    delete m;
    return SQGR_OK;
}

int sqgr_autocorr_create_cols(sqgr_ctx* ctx, const sqgr_graph* g, const sqgr_matrix* m, int64_t col0, int64_t G, sqgr_autocorr** out) {
    SQGR_REQUIRE(ctx && g && m && out, "null argument");
    SQGR_REQUIRE(m->ctx == ctx, "matrix belongs to a different context");
    SQGR_REQUIRE(m->n_rows == g->n, "matrix has %lld rows, the graph %lld", (long long)m->n_rows, (long long)g->n);
    SQGR_REQUIRE(col0 >= 0 && G >= 1 && col0 + G <= m->n_cols, "columns [%lld, %lld) outside the matrix (%lld columns)", (long long)col0,
                 (long long)(col0 + G), (long long)m->n_cols);
    return autocorr_create(ctx, g, nullptr, G, true, out, m, col0);
}

int sqgr_autocorr_create_colidx(sqgr_ctx* ctx, const sqgr_graph* g, const sqgr_matrix* m, const int32_t* cols, int64_t G,
                                sqgr_autocorr** out) {
    SQGR_REQUIRE(ctx && g && m && cols && out, "null argument");
    SQGR_REQUIRE(m->ctx == ctx, "matrix belongs to a different context");
    SQGR_REQUIRE(m->n_rows == g->n, "matrix has %lld rows, the graph %lld", (long long)m->n_rows, (long long)g->n);
    SQGR_REQUIRE(G >= 1, "G=%lld", (long long)G);
    for (int64_t k = 0; k < G; ++k)
        SQGR_REQUIRE(cols[k] >= 0 && cols[k] < m->n_cols, "cols[%lld]=%d outside the matrix (%lld columns)", (long long)k, cols[k], (long long)m->n_cols);
    SQGR_HIP(hipSetDevice(ctx->device));
    SQGR_TRY(m->ensure_by_column());
    DevBuf<int32_t> d_cols;
    SQGR_TRY(d_cols.alloc((size_t)G));
    SQGR_HIP(hipMemcpyAsync(d_cols.p, cols, (size_t)G * 4, hipMemcpyHostToDevice, ctx->stream));
    return autocorr_create(ctx, g, nullptr, G, true, out, m, 0, d_cols.p);  // synchronous on return: d_cols may go
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
    SQGR_TRY(h->ensure_stat());
    k_scores<<<(unsigned)ceil_div(h->G, 256), 256, 0, st>>>(mode, h->G, h->n, h->W, h->tmpG.p, h->z2ss.p, h->isconst.p, h->stat_dev.p);
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipMemcpyAsync(h->stat_host, h->stat_dev.p, (size_t)h->G * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    memcpy(out_scores, h->stat_host, (size_t)h->G * 8);
    return SQGR_OK;
}

// permutation scores for permutations [perm_begin, perm_end): row permutations injected from the host (perm_idx), drawn
// from numpy's streams on the device (pcg_states, one row per permutation of the range) or from the device generator
static int autocorr_perms(sqgr_autocorr* h, int32_t mode, const int32_t* perm_idx, const uint64_t* pcg_states, uint64_t seed,
                          int64_t perm_begin, int64_t perm_end, double* out_sims, double* dev_all = nullptr) {
    SQGR_REQUIRE(h && (out_sims || dev_all), "null argument");
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
    const int rmode = lds_rmode(h, mode);
    const bool geary_lds = rmode == 1;  // layout of the LDS kernel (chunk length): only the row-sum-array variant differs from Moran's
    const int kernel = perm_kernel_choice(n, G, P, geary_lds);
    const bool use_lds = kernel != 0;
    const int split = kernel == 2 ? LDS_SPLIT : 1;
    int64_t by_part = std::max<int64_t>(PERM_TILE, ((int64_t)1 << 30) / ((int64_t)h->ntiles * R * GT * 8));
    if (use_lds) {
        int nch = 0;
        (void)lds_chunk(n, geary_lds, &nch);
        by_part = std::max<int64_t>(64, ((int64_t)1 << 30) / (((G + 1) / 2) * nch * GP * 8 * split));
    }
    // The joint list schedule (k_bucket_order_joint) makes a permutation's summation order depend on the 15 permutations it shares a
    // `ds_read_b128` lane group with, so the LDS kernel works on whole 16-aligned groups of the GLOBAL permutation index: with the
    // device generator the range is widened to [begin - lead, end rounded up to 16) and the extra ("ghost") permutations are
    // generated, scheduled, scored and dropped — INSIDE ONE KERNEL a permutation's score does not depend on how a range was cut.
    // (The kernel itself is chosen from the length of the call's range, perm_kernel_choice: pieces on different sides of its
    // thresholds sum in different orders — include/sqgr.h says so; the front end always passes the whole range.  Injected
    // permutations and numpy's streams always start at permutation 0.)
    const bool ghosts = use_lds && !perm_idx && !pcg_states;
    const int64_t galign = 16 / split;  // (perm_slot: a service group holds 16 consecutive lanes = permutations, or 2 x 8 sub-lists)
    const int64_t lead = ghosts ? (perm_begin & (galign - 1)) : 0;
    const int64_t PV = ghosts ? ((lead + P + galign - 1) & ~(galign - 1)) : P;  // permutations the passes run over
    int64_t chunk = std::min<int64_t>(std::min<int64_t>(PV, 32768), std::min(by_idx, by_part));  // grid.y limit
    const int64_t wg_perms = LDS_PERM_BLOCK / split;  // permutations per workgroup of the LDS kernel
    if (use_lds && chunk > wg_perms) chunk = chunk / wg_perms * wg_perms;  // whole workgroups of permutations
    else if (use_lds && chunk < PV) chunk = std::max<int64_t>(galign, chunk / galign * galign);  // every pass starts a lane group
    SQGR_TRY(h->idx.ensure((size_t)chunk * n));
    if (!use_lds) {
        SQGR_TRY(h->part1.ensure((size_t)h->ntiles * chunk * R * GT));
        if (mode == 1) SQGR_TRY(h->part2.ensure((size_t)h->ntiles * chunk * R * GT));
    }
    SQGR_TRY(h->sims.ensure((size_t)chunk * G));
    if (pcg_states) SQGR_TRY(h->pcg_states.ensure((size_t)chunk * 4));
    if (perm_idx)
        for (int64_t t = 0; t < P * n; ++t)
            SQGR_REQUIRE(perm_idx[t] >= 0 && perm_idx[t] < n, "perm_idx[%lld]=%d outside [0,%lld)", (long long)t, perm_idx[t], (long long)n);
    const FeistelDomain dom = make_domain((uint32_t)n);
    for (int64_t c0 = 0; c0 < PV; c0 += chunk) {
        const int64_t pc = std::min(chunk, PV - c0);
        const int64_t real_lo = std::max(c0, lead), real_hi = std::min(c0 + pc, lead + P);  // this pass's part of the caller's range
        // the LDS kernel walks bucket lists that depend on the permutations alone: reuse the context's if they are these
        PermLists* pl = use_lds ? perm_lists(ctx) : nullptr;
        int lm = 0, lnch = 0;
        if (use_lds) lm = lds_chunk(n, geary_lds, &lnch);
        const int64_t perm0 = (perm_idx || pcg_states) ? c0 : perm_begin - lead + c0;
        const int kind = perm_idx ? 2 : (pcg_states ? 1 : 0);
        // lists with row-sum classes in their entries serve RMODE 2 only; the order of the lists is part of what they are
        // (1 << 11: with exception lists, RMODE 3)
        const int list_kind = split | (rmode == 2 ? 256 : 0) | (list_order_mode(split) << 9) | (rmode == 3 ? 2048 : 0);
        const uint64_t lists_cls = rmode >= 2 ? h->cls_serial : 0;
        const bool hit = use_lds && pl->matches(n, pc, perm0, lm, lnch, kind, seed, pcg_states ? pcg_states + (size_t)c0 * 4 : nullptr, list_kind,
                                             lists_cls);
        if (!hit) {
            if (perm_idx) {
                SQGR_HIP(hipMemcpyAsync(h->idx.p, perm_idx + (size_t)c0 * n, (size_t)pc * n * 4, hipMemcpyHostToDevice, st));
            } else if (pcg_states) {
                SQGR_HIP(hipMemcpyAsync(h->pcg_states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
                SQGR_TRY(pcg_permutations_dev(ctx, h->pcg_ws, n, h->pcg_states.p, pc, h->idx.p, st));
            } else {
                LaunchTimer t(ctx, "autocorr_perm_indices");
                k_perm_indices<<<dim3((unsigned)ceil_div(n, 256), (unsigned)pc), 256, 0, st>>>(seed, perm_begin - lead + c0, n, dom, h->idx.p);
                SQGR_HIP(hipGetLastError());
            }
        }
        if (use_lds) {
            if (!hit) {
                SQGR_TRY(build_perm_lists(ctx, pl, h->idx.p, n, pc * split, perm0, lm, lnch, split, rmode != 0, rmode == 2 ? h->rcls.p : nullptr,
                                          rmode == 3 ? h->rcls.p : nullptr, h->rs_main));
                pl->n = n; pl->pc = pc; pl->perm0 = perm0; pl->m = lm; pl->nch = lnch; pl->kind = kind; pl->seed = seed; pl->split = list_kind;
                pl->cls_serial = lists_cls;
                pl->states.clear();
                if (pcg_states) pl->states.assign(pcg_states + (size_t)c0 * 4, pcg_states + (size_t)(c0 + pc) * 4);
            }
            SQGR_TRY(perms_pass_lds(h, mode, pc, pl, split));
            if (real_hi > real_lo) {
                const double* from = h->sims.p + (size_t)(real_lo - c0) * G;
                const size_t to = (size_t)(real_lo - lead) * G, bytes = (size_t)(real_hi - real_lo) * G * 8;
                if (dev_all) SQGR_HIP(hipMemcpyAsync(dev_all + to, from, bytes, hipMemcpyDeviceToDevice, st));
                if (out_sims) SQGR_HIP(hipMemcpyAsync(out_sims + to, from, bytes, hipMemcpyDeviceToHost, st));
            }
            SQGR_HIP(hipStreamSynchronize(st));
            continue;
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
        if (dev_all) SQGR_HIP(hipMemcpyAsync(dev_all + (size_t)c0 * G, h->sims.p, (size_t)pc * G * 8, hipMemcpyDeviceToDevice, st));
        if (out_sims) SQGR_HIP(hipMemcpyAsync(out_sims + (size_t)c0 * G, h->sims.p, (size_t)pc * G * 8, hipMemcpyDeviceToHost, st));
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

int sqgr_autocorr_perm_stats(sqgr_autocorr* h, int32_t mode, const int32_t* perm_idx, const uint64_t* pcg_states, uint64_t seed,
                             int64_t perm_begin, int64_t perm_end, const double* score, int64_t* out_ge, double* out_sum,
                             double* out_std, double* out_var, int32_t only_feature) {
    SQGR_REQUIRE(h && score && out_ge && out_sum && out_std && out_var, "null argument");
    SQGR_REQUIRE(!only_feature || h->G == 1, "only_feature is set for a block of %lld features", (long long)h->G);
    SQGR_REQUIRE(!(perm_idx && pcg_states), "perm_idx and pcg_states are mutually exclusive");
    SQGR_REQUIRE(perm_begin >= 0 && perm_end > perm_begin, "bad permutation range");
    sqgr_ctx* ctx = h->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t G = h->G, P = perm_end - perm_begin;
    SQGR_TRY(h->sims_all.ensure((size_t)P * G));
    SQGR_TRY(autocorr_perms(h, mode, perm_idx, pcg_states, seed, (perm_idx || pcg_states) ? 0 : perm_begin, (perm_idx || pcg_states) ? P : perm_end,
                            nullptr, h->sims_all.p));
    SQGR_TRY(h->ensure_stat());
    double* d = h->stat_dev.p;  // [score | sum | std | var | n_ge]
    static_assert(sizeof(long long) == sizeof(double), "the n_ge column shares the block");
    memcpy(h->stat_host, score, (size_t)G * 8);
    SQGR_HIP(hipMemcpyAsync(d, h->stat_host, (size_t)G * 8, hipMemcpyHostToDevice, st));
    {
        LaunchTimer t(ctx, "autocorr_perm_stats");
        k_perm_stats<<<(unsigned)ceil_div(G, 256), 256, 0, st>>>(h->sims_all.p, P, G, d, reinterpret_cast<long long*>(d + 4 * G), d + G, d + 2 * G, d + 3 * G,
                                                                 only_feature ? 1 : 0);
        SQGR_HIP(hipGetLastError());
    }
    SQGR_HIP(hipMemcpyAsync(h->stat_host + G, d + G, (size_t)4 * G * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    memcpy(out_sum, h->stat_host + G, (size_t)G * 8);
    memcpy(out_std, h->stat_host + 2 * G, (size_t)G * 8);
    memcpy(out_var, h->stat_host + 3 * G, (size_t)G * 8);
    memcpy(out_ge, h->stat_host + 4 * G, (size_t)G * 8);
    return SQGR_OK;
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
