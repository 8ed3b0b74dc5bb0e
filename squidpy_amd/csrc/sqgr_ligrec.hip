// libsqgr: ligand-receptor permutation test (CellPhoneDB-style), SURVEY.md §8(f) row 4.
//
// Reference semantics (scverse/squidpy, src/squidpy/gr/_ligrec.py):
//   :616-673  `_score_permutations`: per permutation p
//               perm   = shuffle(clustering)                               (one generator per permutation)
//               groups[k, g] = (sum_{cell: perm[cell]=k} data[cell, g]) * inv_counts[k]   (cells added in index order)
//               counts[i, j] += valid[i, j] and groups[a_j, rec_i] + groups[b_j, lig_i] > mean_obs[a_j, rec_i] + mean_obs[b_j, lig_i]
//   :677-775  `_analysis`: builds mean_obs / mask / valid, pvalues = counts / n_perms
//
// MI355X design.  The expression matrix is kept as CSC columns of its non-zeros (a zero adds nothing to a group sum).
//   k_ligrec_sums : one wave per (gene, 64 permutations).  Lane = permutation.  The wave walks the gene's non-zero
//                   cells in index order; per cell all 64 lanes read their permutation's label of that cell (one
//                   64-byte row of the label matrix, coalesced) and add the value to a private LDS column
//                   acc[label][lane] (ds_add_f64, no bank conflicts, never contended).  Because a lane's adds reach
//                   LDS in program order, every group sum is accumulated in exactly the reference's order — the
//                   f64 result is bit-identical to the sequential CPU loop.  Written out as means[g][k][perm].
//   k_ligrec_score: block per (interaction, 256 cluster pairs).  Stages the two genes' [K][32 perms] mean tiles in
//                   LDS (row stride 33 doubles: the K rows start in distinct banks), thread = cluster pair, loops
//                   the permutations of the tile and counts `shuf > obs` in a register; one owner per output cell,
//                   no atomics.
// Labels come from the generators of sqgr_nhood.hip (sqgr_shuffle.h): Philox-keyed Feistel (default) or numpy's
// PCG64 streams (bit-for-bit `Generator.shuffle`).
#include "sqgr_shuffle.h"

#include <algorithm>

namespace sqgr {

constexpr int SCORE_TP = 32;        // permutations per LDS tile of the score kernel
constexpr int SCORE_LD = SCORE_TP + 1;

// label of `cell` in permutation q of this launch: labels[(q / seg) * seg_stride + cell * row_stride + q % seg]
struct LabelView {
    const uint8_t* base;
    int64_t row_stride;
    int64_t seg_stride;
    int seg;
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_ligrec_sums(int G, int K, const int64_t* __restrict__ colptr,
                                                            const int32_t* __restrict__ rowidx, const double* __restrict__ vals,
                                                            LabelView lv, const double* __restrict__ inv_counts, int64_t npl,
                                                            int k_first, int k_out, int k_off, double* __restrict__ means) {
    extern __shared__ double s_acc[];  // [WAVES][K][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x * WAVES + wave;
    if (g >= G) return;
    double* acc = s_acc + (size_t)wave * K * 64;
    for (int k = 0; k < K; ++k) acc[k * 64 + lane] = 0.0;
    const int64_t q = (int64_t)blockIdx.y * 64 + lane;
    const uint8_t* lab_lane = lv.base + (q / lv.seg) * lv.seg_stride + (q % lv.seg);
    const int64_t e_begin = colptr[g], e_end = colptr[g + 1];
    for (int64_t e0 = e_begin; e0 < e_end; e0 += 64) {
        // 64 non-zeros of the column per trip: lane l holds (cell, value) of entry e0 + l; padding adds +0.0 at cell 0
        const bool in = e0 + lane < e_end;
        const int my_cell = in ? rowidx[e0 + lane] : 0;
        const double my_val = in ? vals[e0 + lane] : 0.0;
        const int my_lo = __double2loint(my_val), my_hi = __double2hiint(my_val);
        const int cnt = (int)((e_end - e0 < 64) ? e_end - e0 : 64);
        constexpr int GU = 16;  // label rows in flight per wave: the kernel runs at 8 waves per CU (LDS), latency is hidden
                                // by memory-level parallelism inside the wave
        const int cntu = (cnt + GU - 1) & ~(GU - 1);
        for (int j0 = 0; j0 < cntu; j0 += GU) {
            uint32_t lab[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int cell = __builtin_amdgcn_readlane(my_cell, j0 + u);
                lab[u] = lab_lane[(int64_t)cell * lv.row_stride];
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const double v = __hiloint2double(__builtin_amdgcn_readlane(my_hi, j0 + u), __builtin_amdgcn_readlane(my_lo, j0 + u));
                // entries past `cnt` were padded with +0.0: the add is a no-op
                __hip_atomic_fetch_add(&acc[lab[u] * 64 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    // the wave's own LDS atomics complete in order; wait for them before reading back
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (cluster tiles: local label k >= k_first is cluster k_off + k - k_first of k_out; local label 0 collects the other tiles' cells)
    double* out = means + ((size_t)g * k_out + k_off) * npl + q;
    for (int k = k_first; k < K; ++k) out[(size_t)(k - k_first) * npl] = acc[k * 64 + lane] * inv_counts[k];
}

__global__ __launch_bounds__(256) void k_ligrec_score(int K, int n_cp, const int32_t* __restrict__ inter,
                                                      const int32_t* __restrict__ cpairs, const double* __restrict__ obs,
                                                      const uint8_t* __restrict__ valid, const double* __restrict__ means,
                                                      int64_t npl, int first_valid, int n_valid_perms,
                                                      int64_t* __restrict__ counts) {
    extern __shared__ double s_tile[];  // [2][K][SCORE_LD]
    double* s_rec = s_tile;
    double* s_lig = s_tile + (size_t)K * SCORE_LD;
    const int i = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;
    const int rec = inter[2 * i], lig = inter[2 * i + 1];
    const bool live = j < n_cp;
    const int a = live ? cpairs[2 * j] : 0, b = live ? cpairs[2 * j + 1] : 0;
    const double o = live ? obs[(size_t)i * n_cp + j] : 0.0;
    const double* m_rec = means + (size_t)rec * K * npl;
    const double* m_lig = means + (size_t)lig * K * npl;
    int cnt = 0;
    for (int p0 = 0; p0 < n_valid_perms; p0 += SCORE_TP) {
        __syncthreads();
        for (int t = threadIdx.x; t < K * SCORE_TP; t += 256) {
            const int k = t / SCORE_TP, pl = t % SCORE_TP;
            s_rec[k * SCORE_LD + pl] = m_rec[(size_t)k * npl + p0 + pl];
            s_lig[k * SCORE_LD + pl] = m_lig[(size_t)k * npl + p0 + pl];
        }
        __syncthreads();
        const int np = (n_valid_perms - p0 < SCORE_TP) ? n_valid_perms - p0 : SCORE_TP;
        const double* ra = s_rec + a * SCORE_LD;
        const double* lb = s_lig + b * SCORE_LD;
        // columns before `first_valid` belong to permutations in front of the requested range (ranges start inside a
        // 16-permutation group of the label generator: the whole group is generated)
        for (int pl = (first_valid > p0 ? first_valid - p0 : 0); pl < np; ++pl) cnt += (ra[pl] + lb[pl] > o) ? 1 : 0;
    }
    if (live && valid[(size_t)i * n_cp + j]) counts[(size_t)i * n_cp + j] += cnt;
}

// More than 256 clusters: the clusters are processed in tiles of at most 255.  labels8[cell * npl + q] = 1 + (l - k0) for a
// 16-bit label l of the tile [k0, k1), 0 for every other cluster; slab16[(q / 16 * n + cell) * 16 + q % 16] from the generators.
__global__ __launch_bounds__(256) void k_ligrec_tile_labels(int64_t n, int64_t npl, const uint16_t* __restrict__ slab16, int k0, int k1,
                                                            uint8_t* __restrict__ labels8) {
    const int64_t cell = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (cell >= n) return;
    const int batch = blockIdx.y;
    const uint4* src = reinterpret_cast<const uint4*>(slab16 + ((size_t)batch * n + cell) * 16);
    const uint4 lo = src[0], hi = src[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int l = (int)((w[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu);
        const uint32_t v = (l >= k0 && l < k1) ? (uint32_t)(l - k0 + 1) : 0u;
        out[b >> 2] |= v << ((b & 3) * 8);
    }
    *reinterpret_cast<uint4*>(labels8 + (size_t)cell * npl + (size_t)batch * 16) = make_uint4(out[0], out[1], out[2], out[3]);
}

// the score kernel without the LDS tiles (2 * K * 33 float64 do not fit beyond K = 310): every thread streams the rows of its
// own cluster pair
__global__ __launch_bounds__(256) void k_ligrec_score_direct(int K, int n_cp, const int32_t* __restrict__ inter,
                                                             const int32_t* __restrict__ cpairs, const double* __restrict__ obs,
                                                             const uint8_t* __restrict__ valid, const double* __restrict__ means,
                                                             int64_t npl, int first_valid, int n_valid_perms,
                                                             int64_t* __restrict__ counts) {
    const int i = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= n_cp || !valid[(size_t)i * n_cp + j]) return;
    const int rec = inter[2 * i], lig = inter[2 * i + 1];
    const double o = obs[(size_t)i * n_cp + j];
    const double* ra = means + ((size_t)rec * K + cpairs[2 * j]) * npl;
    const double* lb = means + ((size_t)lig * K + cpairs[2 * j + 1]) * npl;
    int cnt = 0;
    for (int pl = first_valid; pl < n_valid_perms; ++pl) cnt += (ra[pl] + lb[pl] > o) ? 1 : 0;
    counts[(size_t)i * n_cp + j] += cnt;
}

template <typename KernelT>
static int allow_lds(KernelT kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return SQGR_OK;
    SQGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SQGR_OK;
}

struct ShufflerGuard {
    LabelShuffler* s = nullptr;
    ~ShufflerGuard() {
        if (s) label_shuffler_destroy(s);
    }
};

}  // namespace sqgr

using namespace sqgr;

extern "C" {

int sqgr_ligrec_counts(sqgr_ctx* ctx, int64_t n_cells, int32_t n_genes, int32_t K, const int64_t* colptr, const int32_t* rowidx,
                       const double* values, const int32_t* clustering, const double* inv_counts, const int32_t* interactions,
                       int64_t n_inter, const int32_t* cpairs, int32_t n_cp, const double* obs, const uint8_t* valid,
                       uint64_t seed, const uint64_t* pcg_states, int64_t perm_begin, int64_t perm_end, int64_t* out_counts,
                       double* out_means_perm0) {
    SQGR_REQUIRE(ctx && colptr && clustering && inv_counts && interactions && cpairs && obs && valid && out_counts,
                 "null argument");
    SQGR_REQUIRE(n_cells > 0 && n_genes > 0 && n_inter > 0 && n_cp > 0, "empty problem (n_cells=%lld, n_genes=%d, n_inter=%lld, n_cp=%d)",
                 (long long)n_cells, n_genes, (long long)n_inter, n_cp);
    SQGR_REQUIRE(perm_begin >= 0 && perm_end >= perm_begin, "bad permutation range [%lld,%lld)", (long long)perm_begin,
                 (long long)perm_end);
    SQGR_REQUIRE(n_inter < ((int64_t)1 << 31), "too many interactions");
    const int64_t nnz = colptr[n_genes];
    SQGR_REQUIRE(colptr[0] == 0 && nnz >= 0 && (nnz == 0 || (rowidx && values)), "malformed CSC arrays");
    for (int g = 0; g < n_genes; ++g) SQGR_REQUIRE(colptr[g + 1] >= colptr[g], "colptr is not non-decreasing at %d", g);
    for (int64_t e = 0; e < nnz; ++e)
        SQGR_REQUIRE(rowidx[e] >= 0 && rowidx[e] < n_cells, "rowidx[%lld]=%d outside [0,%lld)", (long long)e, rowidx[e], (long long)n_cells);
    for (int64_t i = 0; i < 2 * n_inter; ++i)
        SQGR_REQUIRE(interactions[i] >= 0 && interactions[i] < n_genes, "interaction gene id %d outside [0,%d)", interactions[i], n_genes);
    for (int64_t j = 0; j < 2 * (int64_t)n_cp; ++j)
        SQGR_REQUIRE(cpairs[j] >= 0 && cpairs[j] < K, "cluster id %d outside [0,%d)", cpairs[j], K);
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ShufflerGuard sh;
    SQGR_TRY(label_shuffler_create(ctx, n_cells, clustering, K, &sh.s));  // validates the labels, K in [2, 65535]
    // more than 256 clusters: 16-bit labels from the generators, the group sums in cluster tiles of at most 255 (+ one bucket
    // for the cells of the other tiles), the same permutation behind every tile
    const bool wide = label_shuffler_wide(sh.s);
    const int n_tiles = wide ? (int)ceil_div(K, 255) : 1;
    const int tile = wide ? (int)ceil_div(K, n_tiles) : K;
    const int Kt = wide ? tile + 1 : K;  // labels the sum kernel sees

    const size_t n_out = (size_t)n_inter * n_cp;
    DevBuf<int64_t> d_colptr, d_counts;
    DevBuf<int32_t> d_rowidx, d_inter, d_cpairs;
    DevBuf<double> d_vals, d_inv, d_obs, d_means;
    DevBuf<uint8_t> d_valid, d_labels;
    DevBuf<uint16_t> d_slab16;
    DevBuf<uint32_t> d_keys;
    DevBuf<uint64_t> d_states;
    SQGR_TRY(d_colptr.alloc((size_t)n_genes + 1));
    SQGR_TRY(d_rowidx.alloc((size_t)nnz));
    SQGR_TRY(d_vals.alloc((size_t)nnz));
    SQGR_TRY(d_inv.alloc((size_t)K));
    SQGR_TRY(d_inter.alloc((size_t)n_inter * 2));
    SQGR_TRY(d_cpairs.alloc((size_t)n_cp * 2));
    SQGR_TRY(d_obs.alloc(n_out));
    SQGR_TRY(d_valid.alloc(n_out));
    SQGR_TRY(d_counts.alloc(n_out));
    SQGR_HIP(hipMemcpyAsync(d_colptr.p, colptr, ((size_t)n_genes + 1) * 8, hipMemcpyHostToDevice, st));
    if (nnz) {
        SQGR_HIP(hipMemcpyAsync(d_rowidx.p, rowidx, (size_t)nnz * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(d_vals.p, values, (size_t)nnz * 8, hipMemcpyHostToDevice, st));
    }
    SQGR_HIP(hipMemcpyAsync(d_inv.p, inv_counts, (size_t)K * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(d_inter.p, interactions, (size_t)n_inter * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(d_cpairs.p, cpairs, (size_t)n_cp * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(d_obs.p, obs, n_out * 8, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemcpyAsync(d_valid.p, valid, n_out, hipMemcpyHostToDevice, st));
    SQGR_HIP(hipMemsetAsync(d_counts.p, 0, n_out * 8, st));

    // permutations per launch group: labels take n_cells bytes, the group means G*K*8 bytes per permutation
    // the device generator works in groups of 16 permutations (sqgr_rng.h): start at the group boundary at or below
    // perm_begin and leave the `skip` columns in front of the range out of the scores
    const int64_t skip = pcg_states ? 0 : (perm_begin % 16);
    const int64_t n_perms = perm_end - perm_begin + skip;
    size_t free_b = 0, total_b = 0;
    SQGR_HIP(hipMemGetInfo(&free_b, &total_b));
    const int64_t per_perm = n_cells * (wide ? (pcg_states ? 7 : 3) : 1) + (int64_t)n_genes * K * 8;
    int64_t npl = (int64_t)std::min<size_t>(free_b / 4, (size_t)32 << 30) / per_perm / 64 * 64;
    npl = std::max<int64_t>(64, std::min<int64_t>(npl, 16384));
    npl = std::min<int64_t>(npl, ceil_div(std::max<int64_t>(n_perms, 1), 64) * 64);
    SQGR_TRY(d_means.alloc((size_t)n_genes * K * npl));
    SQGR_TRY(d_labels.alloc((size_t)n_cells * npl));
    if (wide) SQGR_TRY(d_slab16.alloc((size_t)n_cells * npl));
    if (pcg_states)
        SQGR_TRY(d_states.alloc((size_t)npl * 4));
    else
        SQGR_TRY(d_keys.alloc((size_t)npl * 8));

    // waves (= genes) per block of the sum kernel: K*64 doubles of LDS each
    const size_t lds_wave = (size_t)Kt * 64 * 8;
    int waves = (int)std::min<size_t>(4, (160 * 1024) / lds_wave);
    if (waves == 3) waves = 2;
    SQGR_REQUIRE(waves >= 1, "K=%d: the per-wave accumulator does not fit LDS", Kt);
    const size_t lds_sums = lds_wave * waves;
    const size_t lds_score = wide ? 0 : (size_t)2 * K * SCORE_LD * 8;
    if (waves == 4) SQGR_TRY(allow_lds(k_ligrec_sums<4>, lds_sums));
    else if (waves == 2) SQGR_TRY(allow_lds(k_ligrec_sums<2>, lds_sums));
    else SQGR_TRY(allow_lds(k_ligrec_sums<1>, lds_sums));
    if (!wide) SQGR_TRY(allow_lds(k_ligrec_score, lds_score));

    for (int64_t c0 = 0; c0 < n_perms; c0 += npl) {
        const int64_t pc = std::min(npl, n_perms - c0);
        const int64_t pc64 = ceil_div(pc, 64) * 64;
        LabelView lv;
        lv.base = d_labels.p;
        auto launch_sums = [&](int k_local, const double* inv, int k_first, int k_off) -> int {
            LaunchTimer t(ctx, "ligrec_sums");
            dim3 grid((unsigned)ceil_div(n_genes, waves), (unsigned)(pc64 / 64));
            if (waves == 4)
                k_ligrec_sums<4><<<grid, 256, lds_sums, st>>>(n_genes, k_local, d_colptr.p, d_rowidx.p, d_vals.p, lv, inv, npl, k_first, K, k_off, d_means.p);
            else if (waves == 2)
                k_ligrec_sums<2><<<grid, 128, lds_sums, st>>>(n_genes, k_local, d_colptr.p, d_rowidx.p, d_vals.p, lv, inv, npl, k_first, K, k_off, d_means.p);
            else
                k_ligrec_sums<1><<<grid, 64, lds_sums, st>>>(n_genes, k_local, d_colptr.p, d_rowidx.p, d_vals.p, lv, inv, npl, k_first, K, k_off, d_means.p);
            SQGR_HIP(hipGetLastError());
            return SQGR_OK;
        };
        if (wide) {
            if (pcg_states) {
                SQGR_HIP(hipMemcpyAsync(d_states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
                SQGR_TRY(label_shuffler_pcg64_16(sh.s, d_states.p, pc, d_slab16.p, st));
            } else {
                SQGR_TRY(label_shuffler_philox16(sh.s, seed, perm_begin - skip + c0, (int)ceil_div(pc, 16), d_keys.p, d_slab16.p, st));
            }
            lv.row_stride = npl;
            lv.seg = (int)npl;
            lv.seg_stride = 0;
            const int nb16 = (int)ceil_div(pc, 16);
            if ((int64_t)nb16 * 16 < pc64) SQGR_HIP(hipMemsetAsync(d_labels.p, 0, (size_t)n_cells * npl, st));  // columns past the last batch
            for (int k0 = 0; k0 < K; k0 += tile) {
                const int k1 = std::min(K, k0 + tile);
                {
                    LaunchTimer t(ctx, "ligrec_tile_labels");
                    k_ligrec_tile_labels<<<dim3((unsigned)ceil_div(n_cells, 256), (unsigned)nb16), 256, 0, st>>>(n_cells, npl, d_slab16.p, k0, k1,
                                                                                                            d_labels.p);
                    SQGR_HIP(hipGetLastError());
                }
                // (local label l >= 1 is cluster k0 + l - 1: its inverse count sits at d_inv[k0 - 1 + l])
                SQGR_TRY(launch_sums(k1 - k0 + 1, d_inv.p + k0 - 1, 1, k0));
            }
        } else {
            if (pcg_states) {
                SQGR_HIP(hipMemcpyAsync(d_states.p, pcg_states + (size_t)c0 * 4, (size_t)pc * 32, hipMemcpyHostToDevice, st));
                if (pc < pc64) SQGR_HIP(hipMemsetAsync(d_labels.p, 0, (size_t)n_cells * npl, st));  // columns past pc: label 0
                SQGR_TRY(label_shuffler_pcg64(sh.s, d_states.p, pc, npl, d_labels.p, st));
                lv.row_stride = npl;
                lv.seg = (int)npl;
                lv.seg_stride = 0;
            } else {
                SQGR_TRY(label_shuffler_philox(sh.s, seed, perm_begin - skip + c0, (int)(pc64 / 32), d_keys.p, d_labels.p, st));
                lv.row_stride = 32;
                lv.seg = 32;
                lv.seg_stride = n_cells * 32;
            }
            SQGR_TRY(launch_sums(K, d_inv.p, 0, 0));
        }
        if (out_means_perm0 && c0 == 0) {
            // group means of the first permutation of the range, [K][G] like the reference's `groups` (testing hook)
            std::vector<double> col((size_t)n_genes * K);
            SQGR_HIP(hipMemcpy2DAsync(col.data(), 8, d_means.p + skip, (size_t)npl * 8, 8, (size_t)n_genes * K, hipMemcpyDeviceToHost, st));
            SQGR_HIP(hipStreamSynchronize(st));
            for (int g = 0; g < n_genes; ++g)
                for (int k = 0; k < K; ++k) out_means_perm0[(size_t)k * n_genes + g] = col[(size_t)g * K + k];
        }
        {
            LaunchTimer t(ctx, "ligrec_score");
            dim3 grid((unsigned)n_inter, (unsigned)ceil_div(n_cp, 256));
            if (wide)
                k_ligrec_score_direct<<<grid, 256, 0, st>>>(K, n_cp, d_inter.p, d_cpairs.p, d_obs.p, d_valid.p, d_means.p, npl,
                                                            c0 == 0 ? (int)skip : 0, (int)pc, d_counts.p);
            else
                k_ligrec_score<<<grid, 256, lds_score, st>>>(K, n_cp, d_inter.p, d_cpairs.p, d_obs.p, d_valid.p, d_means.p, npl,
                                                             c0 == 0 ? (int)skip : 0, (int)pc, d_counts.p);
            SQGR_HIP(hipGetLastError());
        }
    }
    SQGR_HIP(hipMemcpyAsync(out_counts, d_counts.p, n_out * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

}  // extern "C"
