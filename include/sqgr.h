/*
 * sqgr.h — C ABI of libsqgr.so, the MI355X (gfx950) implementation of Squidpy's `sq.gr`
 * spatial-statistics hot path.
 *
 * The reference (scverse/squidpy) has no FFI on this path: its kernels are numba-JIT Python
 * (SURVEY.md §8b).  Each entry point below therefore replaces a *Python-level* reference
 * function, cited as file:line under /root/reference/src/squidpy; INTEGRATION.md shows the
 * ctypes binding a Squidpy maintainer would add at those call sites.
 *
 * Conventions
 *   - plain C, every function returns 0 (SQGR_OK) or a negative sqgr_status; never throws.
 *     `sqgr_last_error()` returns a thread-local human-readable message for the last failure.
 *   - all array arguments are caller-owned HOST pointers (C-contiguous), copied in/out by the
 *     library and never retained after return, except inside explicit handles
 *     (sqgr_graph, sqgr_nhood, sqgr_points, sqgr_autocorr), which keep DEVICE copies.
 *   - a context owns one device and one HIP stream; a context is not thread-safe, distinct
 *     contexts may be used from distinct threads (one process per GPU is the intended use).
 *   - calls are synchronous on return unless documented otherwise.
 *   - there is NO CPU fallback anywhere in this library: without a usable HIP device
 *     `sqgr_ctx_create` fails.
 */
#ifndef SQGR_H
#define SQGR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQGR_ABI_VERSION 7

typedef enum sqgr_status {
    SQGR_OK = 0,
    SQGR_ERR_INVALID = -1,     /* bad argument (null pointer, out-of-range label, K <= 1, ...) */
    SQGR_ERR_HIP = -2,         /* a HIP runtime call failed; message has the hipError string   */
    SQGR_ERR_NOMEM = -3,       /* device or host allocation failed                               */
    SQGR_ERR_UNSUPPORTED = -4, /* valid request outside what the kernels implement               */
    SQGR_ERR_NODEVICE = -5     /* no HIP device / device index out of range                      */
} sqgr_status;

typedef struct sqgr_ctx sqgr_ctx;
typedef struct sqgr_graph sqgr_graph;
typedef struct sqgr_nhood sqgr_nhood;
typedef struct sqgr_points sqgr_points;
typedef struct sqgr_autocorr sqgr_autocorr;

/* ------------------------------------------------------------------ library / context */
int sqgr_abi_version(void);
const char* sqgr_last_error(void);
int sqgr_device_count(int* out_count);
int sqgr_ctx_create(int device, sqgr_ctx** out_ctx);
int sqgr_ctx_destroy(sqgr_ctx* ctx);
int sqgr_ctx_sync(sqgr_ctx* ctx);
/* Device buffers of 64 MB and more are parked per device when their owner lets go of them and reused by the next request of
 * about that size (bounded by SQGR_POOL_GB, default a quarter of the device memory — the oldest parked buffers go first; other HIP users of the process see parked memory as taken).
 * sqgr_ctx_trim synchronises the device and returns parked buffers of ctx's device to the driver until at most keep_bytes
 * stay parked (0: everything). */
int sqgr_ctx_trim(sqgr_ctx* ctx, int64_t keep_bytes);
/* What the library asked of the HIP allocator since it was loaded (process-wide, all devices): out[0..count) <-
 * {hipMalloc calls, bytes of the successful ones, ns spent inside hipMalloc, hipFree calls, ns spent inside hipFree, requests
 * served by a parked buffer, buffers parked, pool flushes}; slots past 8 read 0.  bench.py prints the differences over its timed
 * regions next to the kernel times, and `a second call allocates nothing` is a test (tests/test_autocorr_gpu.py). */
int sqgr_debug_counters(int64_t* out, int32_t count);
/* name[0..len) <- device name, *cu_count <- compute units, *hbm_bytes <- total device memory */
int sqgr_ctx_device_info(sqgr_ctx* ctx, char* name, int len, int* cu_count, int64_t* hbm_bytes);

/* Per-kernel HIP-event timing on the context's own stream (used by bench.py for the roofline
 * figure).  While enabled every kernel launch is bracketed by hipEventRecord on ctx's stream.
 * sqgr_timer_get sums elapsed ms and launches of all kernels whose name starts with `prefix`. */
int sqgr_timer_enable(sqgr_ctx* ctx, int enable);
int sqgr_timer_reset(sqgr_ctx* ctx);
int sqgr_timer_get(sqgr_ctx* ctx, const char* prefix, double* total_ms, int64_t* launches);
/* writes a ';'-separated list "name:launches:ms" of everything timed so far */
int sqgr_timer_report(sqgr_ctx* ctx, char* buf, int len);

/* ------------------------------------------------------------------ multi-GPU: RCCL communicator
 * One process per GPU.  The path shards by permutation range / row tiles / feature blocks and has ONE exchange: an
 * all-reduce(sum) of exact 64-bit integer accumulators (`Σcount`, `Σcount²`: replaces the reduction the reference's
 * joblib fan-out does by concatenating per-job chunks, gr/_nhood.py:215-231, _utils.py:223-231).  The library binds
 * RCCL itself (librccl.so.1, at the first sqgr_comm_* call); the caller only moves rank 0's unique id to the other
 * ranks (any side channel) — no torch.distributed in the data path.
 *   sqgr_comm_unique_id : rank 0 fills out_id[SQGR_UNIQUE_ID_BYTES] (ncclGetUniqueId)
 *   sqgr_comm_create    : collective over all `world` ranks (ncclCommInitRank) on ctx's device
 *   sqgr_comm_allreduce_i64 : in-place all-reduce of a HOST buffer int64[count] (staged through the device; uint64
 *                             data may be passed through its int64 view for SQGR_OP_SUM: addition modulo 2^64)
 *   sqgr_nhood_set_comm : attaches the communicator to a plan — sqgr_nhood_run then all-reduces the moments ON THE
 *                         DEVICE before its one 2*K*K*8-byte copy-out (every rank returns the global sums), and
 *                         sqgr_nhood_run_pcg64_stats all-gathers the per-permutation counts of the ranks' ranges. */
#define SQGR_UNIQUE_ID_BYTES 128
#define SQGR_OP_SUM 0
#define SQGR_OP_MAX 1
typedef struct sqgr_comm sqgr_comm;
int sqgr_comm_unique_id(uint8_t* out_id);
int sqgr_comm_create(sqgr_ctx* ctx, const uint8_t* unique_id, int32_t rank, int32_t world, sqgr_comm** out_comm);
int sqgr_comm_destroy(sqgr_comm* comm);
int sqgr_comm_info(const sqgr_comm* comm, int32_t* rank, int32_t* world);
int sqgr_comm_allreduce_i64(sqgr_comm* comm, int64_t* buf, int64_t count, int32_t op);
int sqgr_comm_barrier(sqgr_comm* comm);

/* ------------------------------------------------------------------ spatial graph (CSR)
 * Device-resident copy of `adata.obsp[<key>_connectivities]` (scipy CSR): what
 * gr/_nhood.py:194,205 and gr/_ppatterns.py:212 read.  `data` may be NULL (binarised use).
 * indptr: int64[n+1]; indices: int32[nnz]; data: float32[nnz] (sqgr_graph_create) or float64[nnz] (…_f64). */
int sqgr_graph_create(sqgr_ctx* ctx, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                      const float* data, sqgr_graph** out_graph);
/* The same with float64 weights (a float64 `obsp` matrix, or one row-normalised in float64 as gr/_ppatterns.py:212-214
 * does): the device keeps and uses float64 weights in either case — float32 input is widened exactly. */
int sqgr_graph_create_f64(sqgr_ctx* ctx, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                          const double* data, sqgr_graph** out_graph);
int sqgr_graph_destroy(sqgr_graph* g);
/* Observations in no spatial order cost the permutation test's count kernel up to 8x: it gathers the label rows of an edge's two
 * endpoints (no counterpart in the reference, whose loop does not care).  `sqgr_graph_renumbered` builds the twin P A P^T of a graph
 * (structure only, canonical CSR) on the device for `order[new] = old`; `sqgr_spatial_order` computes such an order from coordinates
 * xy float64[n][2] (Z-order curve).  A plan created on the twin with the labels in twin order and `sqgr_nhood_set_spot_map(plan,
 * order)` returns exactly the moments of the plan on the caller's own graph (sqgr_nhood_run; ABI v7). */
int sqgr_graph_renumbered(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* order, sqgr_graph** out_graph);
int sqgr_spatial_order(sqgr_ctx* ctx, const double* xy, int64_t n, int32_t* out_order);

/* ------------------------------------------------------------------ nhood_enrichment
 * replaces the generated numba kernel `_nenrich_{K}_{parallel}` (gr/_nhood.py:54-141):
 *   out[a*K+b] = sum_{i: lab_i=a} #{j in N(i): lab_j=b}      (uint32, edges binarised)
 * labels: int32[n] in [0,K).  2 <= K <= 256. */
int sqgr_nhood_counts(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, uint32_t* out_counts);

/* Injected permutations: the body of `_nhood_enrichment_helper` (gr/_nhood.py:530-539) with the
 * shuffled label vectors supplied by the caller (e.g. numpy's PCG64 shuffles, for exact parity):
 *   labels: uint8[n_perms][n];  out_counts: uint32[n_perms][K][K]. */
int sqgr_nhood_counts_batch(sqgr_ctx* ctx, const sqgr_graph* g, const uint8_t* labels, int64_t n_perms, int32_t K,
                            uint32_t* out_counts);

/* A resident permutation-test problem: graph + base labels (+ optional libraries) on the device.
 * lib_ids: int32[n] in [0,n_libs) or NULL (gr/_utils.py:185-213 `_shuffle_group` semantics:
 * labels are shuffled independently inside each library). */
int sqgr_nhood_create(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, const int32_t* lib_ids,
                      int32_t n_libs, sqgr_nhood** out_plan);
int sqgr_nhood_destroy(sqgr_nhood* plan);

/* Runs permutations [perm_begin, perm_end) of the test entirely on the device: replaces
 * `parallelize(_nhood_enrichment_helper, ...)` (gr/_nhood.py:215-230).  Label shuffles are generated
 * on the GPU by a counter-based generator keyed by (seed, global permutation index, library):
 * Philox4x32-10 round keys + 8-round Feistel bijection with cycle walking (csrc/sqgr_rng.h,
 * restated in oracle/devrng.py), so results do not depend on how a permutation range is split
 * across calls, devices or ranks.
 *   shift:      int64[K*K] or NULL(=0): d = count - shift is what gets accumulated (exact integers)
 *   out_sum:    int64[K*K]   sum_p d           out_sumsq: uint64[K*K]  sum_p d*d
 *   out_perms:  uint32[(perm_end-perm_begin)*K*K] or NULL — the per-permutation counts. */
int sqgr_nhood_run(sqgr_nhood* plan, uint64_t seed, int64_t perm_begin, int64_t perm_end, const int64_t* shift,
                   int64_t* out_sum, uint64_t* out_sumsq, uint32_t* out_perms);

/* The same test with numpy's OWN random streams reproduced bit for bit on the device: permutation p is
 * shuffled by numpy's algorithm (PCG64 + Generator.shuffle's reverse Fisher-Yates with masked rejection; with libraries the
 * per-library sub-shuffles of `_shuffle_group`, gr/_utils.py:185-213) starting from generator state
 *   pcg_states[4*p .. 4*p+3] = {state_hi, state_lo, inc_hi, inc_lo}
 * = `np.random.PCG64(SeedSequence(seed).spawn(n_perms)[p]).state["state"]`, i.e. the state of `generators[p]` of
 * `spawn_generators` (_utils.py:240-241).  Outputs as sqgr_nhood_run; with out_perms the caller can apply the reference's
 * float64 `perms.mean/std` (gr/_nhood.py:231) and obtain Squidpy's z-scores for that seed exactly. */
int sqgr_nhood_run_pcg64(sqgr_nhood* plan, const uint64_t* pcg_states, int64_t n_perms, const int64_t* shift, int64_t* out_sum,
                         uint64_t* out_sumsq, uint32_t* out_perms);
/* The numpy-stream test reduced the way the reference reduces it (gr/_nhood.py:231): out_mean / out_std float64[K*K] are
 * `perms.mean(axis=0)` and `perms.std(axis=0)` of the (n_perms, K, K) float64 count array, bit for bit (sequential
 * float64 accumulation in permutation order, as numpy does over a leading axis), so that
 * `(count - mean) / std` is Squidpy's z-score for the seed — without moving the n_perms*K*K counts to the host. */
int sqgr_nhood_run_pcg64_stats(sqgr_nhood* plan, const uint64_t* pcg_states, int64_t n_perms, double* out_mean, double* out_std);

/* Launch geometry of a plan, int64[12]: {width of the label slab's rows (16|32 permutations), batches per launch group,
 * edge chunks (count blocks) per batch, edges of the list the count kernel walks (the half list r < c + self loops on a structurally
 * symmetric graph, else nnz), 0 full list | 1 half list | 2 half list with self loops, accumulator slots per batch (K*K x row
 * width), self loops, permutations per group of the label generator, permutations per PASS over the edge list (16 | 8 | 4 | 2 | 1;
 * 32 for 32-wide rows; 0: device-scope counters), row halves per pass (2: 203 <= K <= 256 with 32-bit counters), counter mode
 * (0: 32-bit counters on K*K pairs; 1: 16-bit on K*K pairs; 2: 16-bit on the K (K + 1) / 2 unordered pairs of a symmetric
 * graph), bytes of one chunk's partial histograms}.  bench.py derives its per-kernel ceilings from these. */
int sqgr_nhood_info(sqgr_nhood* plan, int64_t* out_info);

/* Attaches (comm != NULL) or detaches an RCCL communicator: see "multi-GPU" above. */
int sqgr_nhood_set_comm(sqgr_nhood* plan, sqgr_comm* comm);
/* The plan lives on a renumbered twin of the caller's graph (sqgr_graph_renumbered): spot_of int32[n], slab row i belongs to the
 * caller's observation spot_of[i] — the device generator permutes THAT observation's rank, so sqgr_nhood_run returns the moments of
 * the plan on the caller's own graph, bit for bit.  No libraries, at most 256 clusters; the numpy-stream and injected-label entry
 * points refuse such a plan (they permute positions).  NULL removes the map. */
int sqgr_nhood_set_spot_map(sqgr_nhood* plan, const int32_t* spot_of);

/* numpy's `Generator.permutation(n)` for n_perms generator states (layout as above), on the device:
 * out_idx int32[n_perms][n] — the row permutations of `_score_helper` (gr/_ppatterns.py:269-271). */
int sqgr_pcg64_permutations(sqgr_ctx* ctx, int64_t n, const uint64_t* pcg_states, int64_t n_perms, int32_t* out_idx);

/* Debug/parity hook: the shuffled label vector of one global permutation index, uint8[n]. */
int sqgr_nhood_shuffled_labels(sqgr_nhood* plan, uint64_t seed, int64_t perm, uint8_t* out_labels);

/* tuning knobs (0 = library default): perms per CSR pass (16|32 = width of the label slab; 8|4|2|1 = cap of the LDS pass width
 * of a 16-wide slab — what 51 <= K <= 202 clusters select by themselves), count blocks (edge chunks) per batch, batches per launch */
int sqgr_nhood_tune(sqgr_nhood* plan, int32_t perms_per_pass, int32_t blocks_per_batch, int32_t batches_per_launch);

/* weighted K x K edge sums: `_interaction_matrix` (gr/_nhood.py:412-429); out: float64[K*K].
 * weights != 0 uses graph data (must have been uploaded), else 1 per stored edge. */
int sqgr_interaction_matrix(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* labels, int32_t K, int32_t weights,
                            double* out);

/* ------------------------------------------------------------------ co_occurrence
 * replaces the numba kernel `_occur_count` (gr/_ppatterns.py:283-310):
 *   out[(a*K+b)*L + r] = #{ i != j : lab_i=a, lab_j=b, d2_ij <= thr2[r] },  d2 = dx*dx + dy*dy in float32
 * x, y: float32[n]; labels: int32[n] in [0,K); thr2: float32[L] squared thresholds (any order).
 * fma = 0: products and sum rounded separately (numpy / literal-source semantics, the oracle of record);
 * fma = 1: d2 = fma(dx,dx,dy*dy) (what LLVM may emit for numba's fastmath=True on an FMA host).
 * Only row tiles t with t % shard_count == shard_index are swept (each unordered tile pair once, credited to
 * both (a,b) and (b,a)); summing the outputs of all shards gives the full counts (multi-GPU: all-reduce). */
int sqgr_cooccur_counts(sqgr_ctx* ctx, const float* x, const float* y, const int32_t* labels, int64_t n, int32_t K,
                        const float* thr2, int32_t L, int32_t fma, int32_t shard_index, int32_t shard_count,
                        int64_t* out_counts);

/* ------------------------------------------------------------------ spatial_autocorr (Moran's I / Geary's C)
 * A resident block of features on one graph: vals float64[G][n] (gene-major, the reference's `vals`,
 * gr/_ppatterns.py:154-185).  The graph must carry its weights (row-normalised by the caller when
 * `transformation=True`, gr/_ppatterns.py:212-214).  mode: 0 = Moran's I, 1 = Geary's C. */
int sqgr_autocorr_create(sqgr_ctx* ctx, const sqgr_graph* g, const double* vals, int64_t G, sqgr_autocorr** out);
/* Same block handed over cell-major: vals float64[n][G] — the layout of `adata.X[:, genes]` before the reference
 * transposes it (gr/_ppatterns.py:168) — so the host never has to materialise the transpose (a strided 8-byte gather,
 * ~0.1 GB/s in numpy: 13 s for 1e5 cells x 2048 genes).  Results are bit-identical to sqgr_autocorr_create. */
int sqgr_autocorr_create_cm(sqgr_ctx* ctx, const sqgr_graph* g, const double* vals, int64_t G, sqgr_autocorr** out);
/* The whole expression matrix resident on the device: x float64[n_rows][n_cols] row-major (`adata.X` as it lies in
 * memory), uploaded once; feature blocks are then columns [col0, col0 + G) of it — no host-side slicing or transposing
 * per block at all (gr/_ppatterns.py:154-185 builds `vals = adata[:, genes].X.T` on the host). */
typedef struct sqgr_matrix sqgr_matrix;
int sqgr_matrix_create(sqgr_ctx* ctx, const double* x, int64_t n_rows, int64_t n_cols, sqgr_matrix** out);
/* The same for float32 (AnnData's usual dtype; widened to float64 on the device, exactly) and / or a column range of a
 * wider host matrix: x points at the first wanted element, rows are `ld` elements apart (ld >= n_cols); value_bytes 4 | 8. */
int sqgr_matrix_create_dense(sqgr_ctx* ctx, const void* x, int32_t value_bytes, int64_t n_rows, int64_t n_cols, int64_t ld,
                             sqgr_matrix** out);
/* The same matrix filled while its first feature blocks are being worked on (ABI v6): `sqgr_matrix_alloc_dense` reserves the
 * device array; `sqgr_matrix_upload_columns(m, x, ld, col0, n_cols)` copies columns [col0, col0 + n_cols) — x points at the first
 * row's element of column col0 of the host matrix, rows `ld` elements apart — on the context's COPY stream and waits for that
 * stream only, so it may run on another host thread than the statistics (config 3: 16 GB cross PCIe in 0.29 s, hidden behind the
 * 0.58 s of the ten feature blocks).  A column range must be uploaded before sqgr_autocorr_create_cols reads it: the caller's
 * business (squidpy_amd: DeviceMatrix.wait_columns). */
int sqgr_matrix_alloc_dense(sqgr_ctx* ctx, int32_t value_bytes, int64_t n_rows, int64_t n_cols, sqgr_matrix** out);
int sqgr_matrix_upload_columns(sqgr_matrix* m, const void* x, int64_t ld, int64_t col0, int64_t n_cols);
/* Sparse expression as scipy holds it — the input every real Visium / Xenium object has (`adata.X` CSR float32; the
 * reference densifies `vals` feature block by feature block on the host, gr/_ppatterns.py:154-185 + scanpy's metrics):
 * CSR (rows = cells: indptr[n_rows + 1], indices = columns) or CSC (columns = features: indptr[n_cols + 1], indices = rows),
 * index arrays int32 or int64 (index_bytes 4 | 8, both arrays alike as in scipy), values float32 or float64
 * (value_bytes 4 | 8), in scipy's canonical format: indices ascending inside every row / column, no repeated entries
 * (checked on the device; SQGR_ERR_INVALID names which — `sort_indices()` / `sum_duplicates()` fix it).  Feature blocks are expanded to dense float64 ON THE DEVICE by sqgr_autocorr_create_cols: the
 * results are bit-identical to uploading `toarray().astype(float64)`. */
int sqgr_matrix_create_csr(sqgr_ctx* ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* indptr, const void* indices,
                           int32_t index_bytes, const void* values, int32_t value_bytes, sqgr_matrix** out);
int sqgr_matrix_create_csc(sqgr_ctx* ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* indptr, const void* indices,
                           int32_t index_bytes, const void* values, int32_t value_bytes, sqgr_matrix** out);
int sqgr_matrix_destroy(sqgr_matrix* m);
int sqgr_autocorr_create_cols(sqgr_ctx* ctx, const sqgr_graph* g, const sqgr_matrix* m, int64_t col0, int64_t G, sqgr_autocorr** out);
/* Features = the columns cols[0 .. G) of the resident matrix, in that order (any subset, any order): the reference's default —
 * the highly variable genes — and explicit `genes` lists select columns with `adata[:, genes].X` on the host
 * (gr/_ppatterns.py:156-166), an O(nnz) copy that costs more than the whole statistic at 1e5 cells; here the matrix is uploaded
 * once as it is and the selection happens on the device (a CSR matrix gets a by-column twin there the first time). */
int sqgr_autocorr_create_colidx(sqgr_ctx* ctx, const sqgr_graph* g, const sqgr_matrix* m, const int32_t* cols, int64_t G,
                                sqgr_autocorr** out);
int sqgr_autocorr_destroy(sqgr_autocorr* h);
/* observed statistic: replaces `score = func(g, vals)` (gr/_ppatterns.py:216; scanpy.metrics.morans_i/gearys_c);
 * constant features -> NaN.  out_scores: float64[G]. */
int sqgr_autocorr_scores(sqgr_autocorr* h, int32_t mode, double* out_scores);
/* permutation scores: replaces `parallelize(_score_helper, ...)` (gr/_ppatterns.py:225-232, 258-280):
 *   out_sims[p][g] = func(g[idx_p, :], vals)[g]        (rows of the graph permuted, values fixed)
 * perm_idx: int32[P][n] caller-supplied permutations (e.g. numpy's `rng.permutation(n)` streams) or NULL to
 * generate permutations [perm_begin, perm_end) on the device (csrc/sqgr_rng.h, keyed by seed and the global
 * permutation index => the PERMUTATIONS are independent of how the range is split).  out_sims: float64[P][G],
 * P = perm_end-perm_begin.  The scores' last bits also depend on which of the three summation kernels runs, and that is
 * chosen from the length of THIS call's range (fewer than 40 permutations, 40-511, 512 and more — for >= 256 features and
 * >= 4096 spots; the gather kernel otherwise): a caller that cuts one test into several calls and wants bit-identical scores
 * keeps every piece on one side of these thresholds (or sets SQGR_AUTOCORR_KERNEL).  The front end never cuts permutation
 * ranges — ranks own feature blocks — so its frames do not depend on the number of ranks. */
int sqgr_autocorr_perms(sqgr_autocorr* h, int32_t mode, const int32_t* perm_idx, uint64_t seed, int64_t perm_begin,
                        int64_t perm_end, double* out_sims);
/* sqgr_autocorr_perms with the reference's own numpy streams generated ON THE DEVICE: pcg_states holds n_perms rows
 * [state_hi, state_lo, inc_hi, inc_lo] of `spawn_generators(seed, n_perms)` (_utils.py:240-241); permutation p of the
 * rows is bit for bit `generators[p].permutation(n)` (gr/_ppatterns.py:270-271).  Nothing but 32 bytes per permutation
 * crosses PCIe (injecting the same permutations through sqgr_autocorr_perms costs 4*n bytes each, per feature block). */
int sqgr_autocorr_perms_pcg64(sqgr_autocorr* h, int32_t mode, const uint64_t* pcg_states, int64_t n_perms, double* out_sims);

/* The permutation test REDUCED on the device: runs the permutations exactly like sqgr_autocorr_perms (perm_idx != NULL:
 * injected; pcg_states != NULL: numpy streams as sqgr_autocorr_perms_pcg64, one row per permutation of the range; both
 * NULL: the device generator for [perm_begin, perm_end)) but keeps the (P, G) scores on the device and returns, per
 * feature, what `_p_value_calc` (gr/_ppatterns.py:474-492) takes out of them — so G x 4 numbers cross PCIe instead of
 * P x G (160 MB at 20 000 features x 1000 permutations):
 *   out_ge[g]  = #{p : sims[p][g] >= score[g]}       `(sims >= score).sum(axis=0)`            int64[G]
 *   out_sum[g] = `sims.sum(axis=0)[g]`    out_std[g] = `sims.std(axis=0)[g]`    out_var[g] = `np.var(sims, axis=0)[g]`
 * each in numpy's own evaluation order (sequential over the permutation axis, mean = sum / P, then the squared
 * deviations summed the same way): bit-identical to numpy on the same scores.  score: float64[G], the observed statistic.
 * only_feature != 0 (G must be 1): this block is the ONLY feature of the call — the reference's (P, 1) array is contiguous
 * along the permutation axis as well and numpy reduces it in its other order (pairwise sums of 8192-element runs); set it
 * when the call scores a single feature, never for a one-feature block of a larger call. */
int sqgr_autocorr_perm_stats(sqgr_autocorr* h, int32_t mode, const int32_t* perm_idx, const uint64_t* pcg_states, uint64_t seed,
                             int64_t perm_begin, int64_t perm_end, const double* score, int64_t* out_ge, double* out_sum,
                             double* out_std, double* out_var, int32_t only_feature);

/* parity hook: the device generator's permutations, int32[perm_end-perm_begin][n] (<= 32768 per call) */
int sqgr_autocorr_perm_indices(sqgr_ctx* ctx, int64_t n, uint64_t seed, int64_t perm_begin, int64_t perm_end,
                               int32_t* out_idx);

/* ------------------------------------------------------------------ ripley (K/L pair counts, F/G kNN distances)
 * metric: 0 euclidean, 1 manhattan, 2 chebyshev (sklearn KDTree arithmetic: per-coordinate accumulation, no FMA); the
 * nearest-neighbour entry points also take 3 canberra (sklearn's BallTree metric without parameters; brute-force sweep,
 * no cell list) — sqgr_pair_counts does not: KDTree.valid_metrics ends at chebyshev (gr/_ripley.py:213).
 *
 * sqgr_pair_counts replaces `KDTree(points).two_point_correlation(points, support, dualtree=True) - m`
 * (gr/_ripley.py:220-222): out[s] = #{ordered i != j : dist_ij <= r_s}.  xy: float64[m][2]; thr: float64[S]
 * ascending.  For metric 0 the comparison is done on the squared distance: pass thr[s] = the largest float64 t with
 * fl(sqrt(t)) <= r_s (so that `d2 <= t` == `sqrt(d2) <= r_s` bit for bit); for metrics 1/2 pass the radii. */
int sqgr_pair_counts(sqgr_ctx* ctx, const double* xy, int64_t m, const double* thr, int32_t S, int32_t metric,
                     int64_t* out_counts);
/* The same for n_sets point sets in ONE launch (Ripley's L evaluates one set per cluster and one per simulation,
 * gr/_ripley.py:152,171): set z = points xy[offsets[z] .. offsets[z+1]), out_counts int64[n_sets][S].  n_sets <= 65535. */
int sqgr_pair_counts_batch(sqgr_ctx* ctx, const double* xy, const int64_t* offsets, int32_t n_sets, const double* thr, int32_t S,
                           int32_t metric, int64_t* out_counts);
/* sqgr_knn_dist replaces `NearestNeighbors(n_neighbors=k).fit(ref).kneighbors(query)[0]` (gr/_ripley.py:144-150,
 * 163-169): out float64[nq][k] ascending; for metric 0 the SQUARED distances (caller applies sqrt). 1 <= k <= 16. */
int sqgr_knn_dist(sqgr_ctx* ctx, const double* query, int64_t nq, const double* ref, int64_t nr, int32_t k, int32_t metric,
                  double* out);
/* Ripley's G keeps its (large) query set on the device: sqgr_points holds coordinates (+ an int32 label per point);
 * sqgr_knn_hist finds, for every query point whose label differs from `exclude_label` (-1: all points), the k nearest of
 * the `ref` points and returns `np.histogram(distances, bins=edges)[0]` (int64[S-1]) of all those distances — what
 * gr/_ripley.py:163-169 + `_f_g_function` :206-209 compute with sklearn and numpy, without moving the distances. */
typedef struct sqgr_points sqgr_points;
int sqgr_points_create(sqgr_ctx* ctx, const double* xy, const int32_t* labels, int64_t n, sqgr_points** out);
int sqgr_points_destroy(sqgr_points* p);
int sqgr_knn_hist(sqgr_ctx* ctx, const sqgr_points* queries, int32_t exclude_label, const double* ref, int64_t nr, int32_t k,
                  int32_t metric, const double* edges, int32_t S, int64_t* out_counts);

/* ------------------------------------------------------------------ spatial graph construction
 * Exact 2-D neighbour search on a device cell list; xy: float64[n][2].
 *
 * sqgr_knn_self replaces `NearestNeighbors(n_neighbors=k, metric="euclidean").fit(xy).kneighbors()`
 * (gr/neighbors.py:196-199, 402-405): for every sample its k nearest OTHER samples (self excluded by index),
 * ascending; ties broken by the smaller index.  out_idx int32[n][k]; out_d2 float64[n][k] SQUARED distances
 * (per-coordinate accumulation, no FMA; the caller applies sqrt).  1 <= k <= min(64, n-1). */
int sqgr_knn_self(sqgr_ctx* ctx, const double* xy, int64_t n, int32_t k, int32_t* out_idx, double* out_d2);
/* sqgr_radius_self replaces `NearestNeighbors(radius=r).fit(xy).radius_neighbors()` (gr/neighbors.py:252-255): all
 * other samples with squared distance <= r*r, as CSR.  Call once with out_idx = out_d2 = NULL to obtain
 * out_indptr int64[n+1], allocate out_indptr[n] entries, call again (capacity = allocated entries). */
int sqgr_radius_self(sqgr_ctx* ctx, const double* xy, int64_t n, double radius, int64_t* out_indptr, int32_t* out_idx,
                     double* out_d2, int64_t capacity);

/* ---- ligand-receptor permutation test -------------------------------------------------------------------------------
 * sqgr_ligrec_counts replaces the numba kernel `_score_permutations` (gr/_ligrec.py:616-673) that `_analysis`
 * (gr/_ligrec.py:677-775) calls once for all permutations:
 *     for every permutation p in [perm_begin, perm_end):
 *         perm        = shuffle(clustering)
 *         groups[k,g] = (sum over cells with perm[cell] == k of data[cell,g], cells in index order) * inv_counts[k]
 *         out_counts[i,j] += valid[i,j] && groups[a_j,rec_i] + groups[b_j,lig_i] > obs[i,j]
 * with (rec_i, lig_i) = interactions[i], (a_j, b_j) = cpairs[j] and obs[i,j] = mean_obs[a_j,rec_i] + mean_obs[b_j,lig_i]
 * (formed by the caller in float64, the same IEEE addition the reference performs per comparison).
 *   data: the (n_cells x n_genes) float64 matrix as CSC columns of its stored entries (colptr int64[n_genes+1],
 *         rowidx int32 ascending inside a column, values) — zeros need not be stored, they do not change a sum.
 *   clustering int32[n_cells] in [0,K), 2 <= K <= 65535 (more than 256 clusters run in cluster tiles of 255 on 16-bit labels); inv_counts float64[K].
 *   pcg_states == NULL: device generator keyed by (seed, global permutation index) — the result for a permutation
 *         range does not depend on how ranges are split over calls or GPUs.
 *   pcg_states != NULL: numpy streams; (perm_end-perm_begin) rows [state_hi,state_lo,inc_hi,inc_lo] of the PCG64
 *         generators `spawn_generators(seed, n_perms)[perm_begin:perm_end]` (_utils.py:240-241); permutation p is then
 *         bit-for-bit `generators[p].shuffle(clustering.copy())`; `seed` is ignored.
 *   out_counts int64[n_inter*n_cp] (row-major, overwritten).  out_means_perm0 (may be NULL): float64[K*n_genes], the
 *         `groups` matrix of the first permutation of the range (for parity checks).
 * Group sums are accumulated in the reference's order (cell index), so they are bit-identical to the CPU loop. */
int sqgr_ligrec_counts(sqgr_ctx* ctx, int64_t n_cells, int32_t n_genes, int32_t K, const int64_t* colptr, const int32_t* rowidx,
                       const double* values, const int32_t* clustering, const double* inv_counts, const int32_t* interactions,
                       int64_t n_inter, const int32_t* cpairs, int32_t n_cp, const double* obs, const uint8_t* valid,
                       uint64_t seed, const uint64_t* pcg_states, int64_t perm_begin, int64_t perm_end, int64_t* out_counts,
                       double* out_means_perm0);

#ifdef __cplusplus
}
#endif
#endif /* SQGR_H */
