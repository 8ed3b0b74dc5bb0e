/*
 * TEST INFRASTRUCTURE — not product code.
 *
 * C restatement of the reference's numba kernels, used (a) as the *timed CPU baseline* of bench.py
 * ("cpu_baseline", kind "port": numba is not installable in this image, so Squidpy's numba/joblib path is
 * timed as this faithful restatement driven by the same numpy RNG calls) and (b) as a second, independent
 * checker next to oracle/restate.py.  Loop structure, scratch arrays and dtypes follow the reference:
 *
 *   sq_nenrich      /root/reference/src/squidpy/gr/_nhood.py:54-141   (res[N,K] uint32 scratch, row scan with
 *                   label gathers, then per-row accumulation into the row label's K-vector)
 *   sq_occur_count  /root/reference/src/squidpy/gr/_ppatterns.py:283-310 (per-point int32[L*K*K] rows, summed)
 *   sq_ligrec_score /root/reference/src/squidpy/gr/_ligrec.py:616-673 (dense groups[K,G] per permutation from the
 *                   shuffled label vector, then the (interaction, cluster pair) indicator loop)
 *   sq_morans_i / sq_gearys_c  scanpy.metrics (third-party; formulas in oracle/restate.py) per-gene loops
 *
 * `parallel` mirrors numba's prange -> `omp parallel for` (the reference default is numba_parallel=False).
 * Compiled without -ffast-math: float32 products and sums are individually rounded, i.e. the arithmetic of the
 * literal source under the shim (the oracle of record).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int sq_nenrich(const uint32_t* indices, const uint32_t* indptr, const uint32_t* clustering, int64_t n, int k,
               uint32_t* out /* k*k */, int parallel) {
    uint32_t* res = (uint32_t*)calloc((size_t)n * k, sizeof(uint32_t)); /* np.zeros((N, K), uint32) */
    if (!res) return -1;
#pragma omp parallel for if (parallel) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint32_t xs = indptr[i], xe = indptr[i + 1];
        for (uint32_t e = xs; e < xe; ++e) res[i * k + clustering[indices[e]]] += 1;
    }
    memset(out, 0, (size_t)k * k * sizeof(uint32_t));
    /* generated if/elif chain `g{cl} += res[row]`; numba turns the prange accumulation into a reduction */
    for (int64_t row = 0; row < n; ++row) {
        uint32_t cl = clustering[row];
        uint32_t* g = out + (size_t)cl * k;
        const uint32_t* r = res + row * k;
        for (int b = 0; b < k; ++b) g[b] += r[b];
    }
    free(res);
    return 0;
}

int sq_occur_count(const float* x, const float* y, const float* thresholds, const int32_t* label_idx, int64_t n, int k,
                   int l_val, int64_t* out /* k*k*l_val */, int parallel) {
    const int64_t k2 = (int64_t)k * k;
    const int64_t width = k2 * l_val;
    int32_t* local = (int32_t*)calloc((size_t)n * width, sizeof(int32_t)); /* np.zeros((n, l_val*k2), int32) */
    if (!local) return -1;
#pragma omp parallel for if (parallel) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t j = 0; j < n; ++j) {
            if (i == j) continue;
            float dx = x[i] - x[j];
            float dy = y[i] - y[j];
            float dx2 = dx * dx, dy2 = dy * dy;
            float d2 = dx2 + dy2;
            int64_t base = ((int64_t)label_idx[i] * k + label_idx[j]) * l_val;
            for (int r = 0; r < l_val; ++r)
                if (d2 <= thresholds[r]) local[i * width + base + r] += 1;
        }
    }
    for (int64_t c = 0; c < width; ++c) out[c] = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t c = 0; c < width; ++c) out[c] += local[i * width + c];
    free(local);
    return 0;
}

/* x: (m, n) row-major float64; weights float64 (already converted like scanpy does) */
int sq_morans_i(const double* data, const int32_t* indices, const int32_t* indptr, const double* x, int64_t m, int64_t n,
                double* out, int parallel) {
    double W = 0.0;
    for (int64_t e = 0; e < indptr[n]; ++e) W += data[e];
#pragma omp parallel for if (parallel) schedule(static)
    for (int64_t g = 0; g < m; ++g) {
        const double* xg = x + g * n;
        double mean = 0.0;
        for (int64_t i = 0; i < n; ++i) mean += xg[i];
        mean /= (double)n;
        double z2 = 0.0, inum = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double zi = xg[i] - mean;
            z2 += zi * zi;
            double acc = 0.0;
            for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) acc += data[e] * (xg[indices[e]] - mean);
            inum += acc * zi;
        }
        out[g] = (double)n / W * inum / z2;
    }
    return 0;
}

int sq_gearys_c(const double* data, const int32_t* indices, const int32_t* indptr, const double* x, int64_t m, int64_t n,
                double* out, int parallel) {
    double W = 0.0;
    for (int64_t e = 0; e < indptr[n]; ++e) W += data[e];
#pragma omp parallel for if (parallel) schedule(static)
    for (int64_t g = 0; g < m; ++g) {
        const double* xg = x + g * n;
        double mean = 0.0;
        for (int64_t i = 0; i < n; ++i) mean += xg[i];
        mean /= (double)n;
        double total = 0.0, den = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
                double d = xg[i] - xg[indices[e]];
                acc += data[e] * d * d;
            }
            total += acc;
            den += (xg[i] - mean) * (xg[i] - mean);
        }
        out[g] = ((double)(n - 1) * total) / (2.0 * W * den);
    }
    return 0;
}

/* `_score_permutations` without the shuffle: perm_labels[n_perms][n_cells] are the shuffled clusterings (drawn by the
 * caller with the same numpy generators).  `parallel` mirrors the prange over permutations. */
int sq_ligrec_score(const double* data, int64_t n_cells, int n_genes, const int32_t* perm_labels, int64_t n_perms, int n_cls,
                    const double* inv_counts, const double* mean_obs, const int32_t* interactions, int64_t n_inter,
                    const int32_t* cpairs, int n_cp, const uint8_t* valid, int64_t* counts, int parallel) {
    memset(counts, 0, (size_t)n_inter * n_cp * sizeof(int64_t));
    int failed = 0;
#pragma omp parallel for if (parallel) schedule(static)
    for (int64_t p = 0; p < n_perms; ++p) {
        const int32_t* perm = perm_labels + p * n_cells;
        double* groups = (double*)calloc((size_t)n_cls * n_genes, sizeof(double));
        if (!groups) {
            failed = 1;
            continue;
        }
        for (int64_t cell = 0; cell < n_cells; ++cell) {
            double* g = groups + (size_t)perm[cell] * n_genes;
            const double* d = data + cell * n_genes;
            for (int j = 0; j < n_genes; ++j) g[j] += d[j];
        }
        for (int k = 0; k < n_cls; ++k)
            for (int j = 0; j < n_genes; ++j) groups[(size_t)k * n_genes + j] *= inv_counts[k];
        for (int64_t i = 0; i < n_inter; ++i) {
            const int rec = interactions[2 * i], lig = interactions[2 * i + 1];
            for (int j = 0; j < n_cp; ++j) {
                if (!valid[i * n_cp + j]) continue;
                const int a = cpairs[2 * j], b = cpairs[2 * j + 1];
                const double shuf = groups[(size_t)a * n_genes + rec] + groups[(size_t)b * n_genes + lig];
                const double obs = mean_obs[(size_t)a * n_genes + rec] + mean_obs[(size_t)b * n_genes + lig];
                if (shuf > obs) {
#pragma omp atomic
                    counts[i * n_cp + j] += 1;
                }
            }
        }
        free(groups);
    }
    return failed ? -1 : 0;
}
