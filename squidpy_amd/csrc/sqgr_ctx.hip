// libsqgr: context, error reporting, kernel timers and the device-resident CSR graph.
#include <dlfcn.h>

#include "sqgr_common.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>

#include <hipcub/hipcub.hpp>

namespace sqgr {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- roctx ranges (declared in sqgr_common.h): libroctx64 is looked up once, only when SQGR_ROCTX=1
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static const RoctxApi& roctx_api() {
    static const RoctxApi api = [] {
        RoctxApi a;
        const char* e = getenv("SQGR_ROCTX");
        if (!(e && atoi(e) == 1)) return a;
        for (const char* nm : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1"}) {
            if (void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) {
                a.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (a.push && a.pop) return a;
                a = RoctxApi();
            }
        }
        return a;
    }();
    return api;
}
void roctx_push(const char* name) {
    if (roctx_api().push) (void)roctx_api().push(name);
}
void roctx_pop() {
    if (roctx_api().pop) (void)roctx_api().pop();
}

// ---- counted device allocations (declared in sqgr_common.h)
AllocStats g_alloc_stats;
static inline int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
hipError_t dev_malloc(void** p, size_t bytes) {
    const int64_t t0 = now_ns();
    const hipError_t e = hipMalloc(p, bytes);
    g_alloc_stats.malloc_ns += now_ns() - t0;
    g_alloc_stats.mallocs += 1;
    if (e == hipSuccess) g_alloc_stats.malloc_bytes += (int64_t)bytes;
    return e;
}
// hipFree waits for the whole device — also for the copy stream of a matrix upload that runs on another host thread
// (sqgr_matrix_alloc_dense ... the last sqgr_matrix_upload_columns): while such an upload is under way the frees are put aside and
// handed to the driver by the first free after it has ended (or by sqgr_ctx_trim / sqgr_ctx_destroy).  Round 6: the ~20 small
// buffers a feature block frees made its kernels wait for the entire 16 GB upload they were meant to overlap.
static std::atomic<int> g_streaming_uploads{0};
static std::mutex g_deferred_mutex;
static std::vector<void*> g_deferred_frees;

void streaming_upload_begin() { g_streaming_uploads += 1; }
void streaming_upload_end() { g_streaming_uploads -= 1; }
static hipError_t free_now(void* p) {
    const int64_t t0 = now_ns();
    const hipError_t e = hipFree(p);
    g_alloc_stats.free_ns += now_ns() - t0;
    g_alloc_stats.frees += 1;
    return e;
}
void flush_deferred_frees() {
    std::vector<void*> todo;
    {
        std::lock_guard<std::mutex> lock(g_deferred_mutex);
        todo.swap(g_deferred_frees);
    }
    for (void* q : todo) (void)free_now(q);
}
hipError_t dev_free(void* p) {
    if (g_streaming_uploads.load() > 0) {
        std::lock_guard<std::mutex> lock(g_deferred_mutex);
        g_deferred_frees.push_back(p);
        return hipSuccess;
    }
    flush_deferred_frees();
    return free_now(p);
}

// ---- parked device buffers (declared in sqgr_common.h)
struct PoolEntry {
    int device;
    void* p;
    size_t cap;
};
static std::mutex g_pool_mutex;
static std::vector<PoolEntry> g_pool;

// Cap of the parked bytes per device: SQGR_POOL_GB, else a quarter of the device's memory (72 GB of the MI355X's 288).  Round 5's
// fixed 32 GB could not hold config 3's 16 GB expression matrix next to what earlier calls had parked: the matrix went back to the
// driver after every call and came back through hipMalloc — which, for memory that has been used before, costs ~30 ms per GB on
// this stack (round 6, counters of bench.py's config-3 leg: 0.7 ms for the first 16 GB of a fresh box, 513 ms for the second).
static size_t pool_limit(int dev) {
    const char* e = getenv("SQGR_POOL_GB");
    if (e && *e) {
        const double gb = atof(e);
        return gb <= 0.0 ? 0 : (size_t)(gb * (double)((size_t)1 << 30));
    }
    static std::map<int, size_t> by_dev;  // (under g_pool_mutex)
    auto it = by_dev.find(dev);
    if (it != by_dev.end()) return it->second;
    size_t total = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) total = prop.totalGlobalMem;
    else (void)hipGetLastError();
    const size_t lim = total ? total / 4 : (size_t)32 << 30;
    by_dev[dev] = lim;
    return lim;
}

struct ComputeStream {
    int device;
    hipStream_t s;
};
static std::mutex g_streams_mutex;
static std::vector<ComputeStream> g_compute_streams;

void pool_register_streams(int device, hipStream_t a, hipStream_t b) {
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    g_compute_streams.push_back({device, a});
    g_compute_streams.push_back({device, b});
}
void pool_unregister_streams(hipStream_t a, hipStream_t b) {
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    for (size_t k = 0; k < g_compute_streams.size();) {
        if (g_compute_streams[k].s == a || g_compute_streams[k].s == b) g_compute_streams.erase(g_compute_streams.begin() + (long)k);
        else ++k;
    }
}
hipError_t pool_quiesce() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::vector<hipStream_t> mine;
    {
        std::lock_guard<std::mutex> lock(g_streams_mutex);
        for (const ComputeStream& c : g_compute_streams)
            if (c.device == dev) mine.push_back(c.s);
    }
    if (mine.empty()) return hipDeviceSynchronize();  // (no context on this device registered: be safe)
    for (hipStream_t s : mine) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void* pool_take(size_t bytes, size_t* capacity) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int best = -1;
    for (int k = 0; k < (int)g_pool.size(); ++k)
        if (g_pool[k].device == dev && g_pool[k].cap >= bytes && g_pool[k].cap - bytes <= bytes / 2 && (best < 0 || g_pool[k].cap < g_pool[best].cap)) best = k;
    if (best < 0) return nullptr;
    void* p = g_pool[best].p;
    *capacity = g_pool[best].cap;
    g_pool.erase(g_pool.begin() + best);
    g_alloc_stats.pool_hits += 1;
    return p;
}

// the device a buffer lives on: the one that allocated it, NOT the one that happens to be current when its owner lets go
static bool owner_device(const void* p, int* dev) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    *dev = attr.device;
    return true;
}

void pool_give(void* p, size_t capacity) {
    int dev = 0;
    if (capacity >= POOL_MIN_BYTES && owner_device(p, &dev)) {
        std::vector<void*> evicted;
        bool parked = false;
        {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            const size_t limit = pool_limit(dev);
            if (capacity <= limit) {
                size_t held = 0;
                for (const PoolEntry& e : g_pool) held += e.device == dev ? e.cap : 0;
                // the newest buffer is the likeliest to be asked for again (the next feature block, the next call): when the cap is
                // reached the OLDEST parked buffers of the device go back to the driver, not the one just released (round 5 refused
                // the newcomer: whatever the first legs of a process had parked stayed, everything later was freed and re-allocated)
                for (size_t k = 0; k < g_pool.size() && held + capacity > limit;) {
                    if (g_pool[k].device != dev) { ++k; continue; }
                    held -= g_pool[k].cap;
                    evicted.push_back(g_pool[k].p);
                    g_pool.erase(g_pool.begin() + (long)k);
                }
                g_pool.push_back({dev, p, capacity});
                g_alloc_stats.pool_parks += 1;
                parked = true;
            }
        }
        for (void* q : evicted) (void)dev_free(q);
        if (parked) return;
    }
    (void)dev_free(p);
}

void pool_trim(int dev, size_t keep_bytes) {  // frees the largest parked buffers of `dev` until at most keep_bytes stay parked
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (;;) {
        size_t held = 0;
        int big = -1;
        for (int k = 0; k < (int)g_pool.size(); ++k) {
            if (g_pool[k].device != dev) continue;
            held += g_pool[k].cap;
            if (big < 0 || g_pool[k].cap > g_pool[big].cap) big = k;
        }
        if (big < 0 || held <= keep_bytes) return;
        (void)dev_free(g_pool[big].p);
        g_pool.erase(g_pool.begin() + big);
    }
}

void pool_flush() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    g_alloc_stats.pool_flushes += 1;
    pool_trim(dev, 0);
}

// one thread per edge: erow[e] = row owning edge e (binary search in indptr; built once per graph)
__global__ void k_expand_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t n, int64_t nnz,
                              int32_t* __restrict__ erow, int2* __restrict__ coo) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int64_t lo = 0, hi = n;  // find largest r with indptr[r] <= e
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= e) lo = mid; else hi = mid;
    }
    erow[e] = (int32_t)lo;
    coo[e] = make_int2((int)((uint32_t)lo * 16u), (int)((uint32_t)indices[e] * 16u));  // byte offsets of 16-byte label rows
}

// ---------------------------------------------------------------------------------------------- renumbered twin of a graph
// The count kernel of the permutation test gathers the label rows of an edge's endpoints: observations in no spatial order cost it
// up to 8x (tools/spot_order_time.py).  sqgr_graph_renumbered builds P A P^T in canonical CSR form on the device for an order given
// as `order[new] = old`; sqgr_spatial_order computes such an order from coordinates (Z-order curve: 16-bit cells, one radix sort).
__global__ void k_order_inverse(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ pos, int* __restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t o = order[i];
    if (o < 0 || o >= n) { flags[0] = 1; return; }
    if (atomicExch(&pos[o], (int32_t)i) != -1) flags[0] = 1;  // an index listed twice: not a permutation
}
__global__ void k_order_degrees(const int64_t* __restrict__ indptr, const int32_t* __restrict__ order, int64_t n, int64_t* __restrict__ deg) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t o = order[i];
    deg[i] = indptr[o + 1] - indptr[o];
}
// one thread per new row: the old row's entries mapped to their new numbers, in ascending order (insertion sort in place: rows of a
// spatial graph hold a handful of entries)
__global__ void k_order_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const int32_t* __restrict__ order,
                             const int32_t* __restrict__ pos, const int64_t* __restrict__ new_indptr, int64_t n, int32_t* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t src = indptr[order[i]], len = indptr[order[i] + 1] - src;
    int32_t* dst = out + new_indptr[i];
    for (int64_t k = 0; k < len; ++k) {
        const int32_t v = pos[indices[src + k]];
        int64_t j = k;
        while (j > 0 && dst[j - 1] > v) {
            dst[j] = dst[j - 1];
            --j;
        }
        dst[j] = v;
    }
}
__global__ void k_morton_keys(const double* __restrict__ xy, int64_t n, double x0, double y0, double sx, double sy, uint32_t* __restrict__ keys,
                              int32_t* __restrict__ idx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto spread = [](uint32_t v) {  // 16 bits -> every second bit of 32
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        return (v | (v << 1)) & 0x55555555u;
    };
    const double fx = (xy[2 * i] - x0) * sx, fy = (xy[2 * i + 1] - y0) * sy;
    const uint32_t qx = (uint32_t)fmin(fmax(fx, 0.0), 65535.0), qy = (uint32_t)fmin(fmax(fy, 0.0), 65535.0);
    keys[i] = spread(qx) | (spread(qy) << 1);
    idx[i] = (int32_t)i;
}

// ---------------------------------------------------------------------------------------------- symmetric half list
// flags[0]: some edge has no mirror; flags[1]: a row is not strictly increasing (unsorted or duplicate entries — the
// binary search below is then meaningless and the graph is treated as not symmetric).
__global__ __launch_bounds__(256) void k_sym_check(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                   const int32_t* __restrict__ erow, int64_t nnz, int* __restrict__ flags) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int32_t r = erow[e], c = indices[e];
    if (e > indptr[r] && indices[e - 1] >= c) flags[1] = 1;
    if (r == c) return;
    int64_t lo = indptr[c], hi = indptr[c + 1];  // first position in row c with index >= r
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (indices[mid] < r) lo = mid + 1; else hi = mid;
    }
    if (lo >= indptr[c + 1] || indices[lo] != r) flags[0] = 1;
}

// class of every edge of a directed graph in canonical form: 1 mutual with r < c (kept: one walk serves the pair), 0 its mirror
// (dropped), 2 no mirror; flags[0]: a self loop exists (the split list is not built then)
__global__ __launch_bounds__(256) void k_split_classify(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                        const int32_t* __restrict__ erow, int64_t nnz, uint8_t* __restrict__ cls,
                                                        int* __restrict__ flags) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int32_t r = erow[e], c = indices[e];
    if (r == c) {
        flags[0] = 1;
        cls[e] = 2;
        return;
    }
    int64_t lo = indptr[c], hi = indptr[c + 1];  // first position in row c with index >= r
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (indices[mid] < r) lo = mid + 1; else hi = mid;
    }
    const bool mutual = lo < indptr[c + 1] && indices[lo] == r;
    cls[e] = mutual ? (r < c ? 1 : 0) : 2;
}

constexpr int HALF_TILE = 1024;  // edges per block of the stable compaction

// pass 1: per tile, the number of edges with r < c and of self loops
// (cls != NULL: the two classes are cls[e] == 1 and cls[e] == 2 instead — sqgr_graph::ensure_split)
__global__ __launch_bounds__(HALF_TILE) void k_half_count(const int2* __restrict__ coo, int64_t nnz, uint32_t* __restrict__ tile_lt,
                                                          uint32_t* __restrict__ tile_self, const uint8_t* __restrict__ cls = nullptr) {
    __shared__ uint32_t s_lt, s_self;
    if (threadIdx.x == 0) s_lt = s_self = 0;
    __syncthreads();
    const int64_t e = blockIdx.x * (int64_t)HALF_TILE + threadIdx.x;
    bool lt = false, self = false;
    if (e < nnz) {
        const int2 rc = coo[e];
        lt = cls ? cls[e] == 1 : rc.x < rc.y;
        self = cls ? cls[e] == 2 : rc.x == rc.y;
    }
    const uint64_t m_lt = __ballot(lt), m_self = __ballot(self);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_lt, (uint32_t)__popcll(m_lt));
        atomicAdd(&s_self, (uint32_t)__popcll(m_self));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_lt[blockIdx.x] = s_lt;
        tile_self[blockIdx.x] = s_self;
    }
}

// pass 2: exclusive scan of the tile counts in place (one block walks the tiles 1024 at a time); totals[0..1] = sums
__global__ __launch_bounds__(1024) void k_half_scan(uint32_t* __restrict__ tile_lt, uint32_t* __restrict__ tile_self, int64_t ntiles,
                                                    unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long s_a[1024], s_b[1024];
    __shared__ unsigned long long carry_a, carry_b;
    if (threadIdx.x == 0) carry_a = carry_b = 0;
    __syncthreads();
    for (int64_t base = 0; base < ntiles; base += 1024) {
        const int64_t t = base + threadIdx.x;
        const unsigned long long va = t < ntiles ? tile_lt[t] : 0, vb = t < ntiles ? tile_self[t] : 0;
        s_a[threadIdx.x] = va;
        s_b[threadIdx.x] = vb;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            unsigned long long xa = 0, xb = 0;
            if ((int)threadIdx.x >= off) {
                xa = s_a[threadIdx.x - off];
                xb = s_b[threadIdx.x - off];
            }
            __syncthreads();
            s_a[threadIdx.x] += xa;
            s_b[threadIdx.x] += xb;
            __syncthreads();
        }
        if (t < ntiles) {  // offsets fit 32 bits: nnz < 2^32 is required by the callers of the half list
            tile_lt[t] = (uint32_t)(carry_a + s_a[threadIdx.x] - va);
            tile_self[t] = (uint32_t)(carry_b + s_b[threadIdx.x] - vb);
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry_a += s_a[1023];
            carry_b += s_b[1023];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = carry_a;
        totals[1] = carry_b;
    }
}

// pass 3: stable scatter — edges with r < c keep their CSR order in out[0, n_lt), self loops follow in out[n_lt, ...)
__global__ __launch_bounds__(HALF_TILE) void k_half_scatter(const int2* __restrict__ coo, int64_t nnz, const uint32_t* __restrict__ tile_lt,
                                                            const uint32_t* __restrict__ tile_self, uint32_t n_lt, int2* __restrict__ out,
                                                            const uint8_t* __restrict__ cls = nullptr) {
    __shared__ uint32_t w_lt[HALF_TILE / 64], w_self[HALF_TILE / 64];
    const int64_t e = blockIdx.x * (int64_t)HALF_TILE + threadIdx.x;
    int2 rc = make_int2(0, 0);
    bool lt = false, self = false;
    if (e < nnz) {
        rc = coo[e];
        lt = cls ? cls[e] == 1 : rc.x < rc.y;
        self = cls ? cls[e] == 2 : rc.x == rc.y;
    }
    const uint64_t m_lt = __ballot(lt), m_self = __ballot(self);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (lane == 0) {
        w_lt[wave] = (uint32_t)__popcll(m_lt);
        w_self[wave] = (uint32_t)__popcll(m_self);
    }
    __syncthreads();
    uint32_t o_lt = tile_lt[blockIdx.x], o_self = tile_self[blockIdx.x];
    for (int w = 0; w < wave; ++w) {
        o_lt += w_lt[w];
        o_self += w_self[w];
    }
    if (lt) out[o_lt + (uint32_t)__popcll(m_lt & below)] = rc;
    if (self) out[(size_t)n_lt + o_self + (uint32_t)__popcll(m_self & below)] = rc;
}

// key of a list entry (16 * row, 16 * col): (row block, column block) of 2^sh spots
__global__ __launch_bounds__(256) void k_tile_keys(const int2* __restrict__ list, int64_t m, int sh, int grid_w, uint64_t* __restrict__ keys) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= m) return;
    const int2 rc = list[e];
    if (grid_w > 0) {  // experiment: 2-D tiles of 2^sh x 2^sh spots of a row-major grid of width grid_w, keyed by the row endpoint
        const uint32_t r = (uint32_t)rc.x >> 4;
        const uint64_t ty = (r / (uint32_t)grid_w) >> sh, tx = (r % (uint32_t)grid_w) >> sh;
        keys[e] = (ty << 31) | tx;
        return;
    }
    const uint64_t rb = ((uint32_t)rc.x >> 4) >> sh, cb = ((uint32_t)rc.y >> 4) >> sh;
    keys[e] = (rb << 31) | cb;
}

}  // namespace sqgr

using namespace sqgr;

// Cache blocking of an edge list of the count kernel (any order gives the same integer counts): the entries are stably sorted
// by (row block, column block) of 2^sh spots, so that consecutive entries gather label rows out of two windows of 2^sh * 16
// bytes — a scan-line order touches every row again one grid line later, long after it left the 32 KB L1.
// SQGR_TILE_EDGES=<sh> (0: off).
static int tile_edge_list(sqgr_ctx* ctx, int2* list, int64_t m, int64_t n) {
    int sh = 0;
    if (const char* env = getenv("SQGR_TILE_EDGES")) sh = atoi(env);
    if (sh <= 0 || m < 2) return SQGR_OK;
    hipStream_t st = ctx->stream;
    DevBuf<uint64_t> keys, keys_out, vals_out;
    SQGR_TRY(keys.alloc((size_t)m));
    SQGR_TRY(keys_out.alloc((size_t)m));
    SQGR_TRY(vals_out.alloc((size_t)m));
    LaunchTimer t(ctx, "graph_tile_edges");
    int grid_w = 0;
    if (const char* env = getenv("SQGR_TILE_GRID_W")) grid_w = atoi(env);
    k_tile_keys<<<(unsigned)ceil_div(m, 256), 256, 0, st>>>(list, m, sh, grid_w, keys.p);
    SQGR_HIP(hipGetLastError());
    (void)n;
    const int bits = 31;  // key = row block << 31 | column block
    size_t tmp_bytes = 0;
    uint64_t* vals_in = reinterpret_cast<uint64_t*>(list);
    SQGR_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_out.p, vals_in, vals_out.p, (int)m, 0, 2 * bits, st));
    DevBuf<uint8_t> tmp;
    SQGR_TRY(tmp.alloc(tmp_bytes));
    SQGR_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys_out.p, vals_in, vals_out.p, (int)m, 0, 2 * bits, st));
    SQGR_HIP(hipMemcpyAsync(list, vals_out.p, (size_t)m * sizeof(int2), hipMemcpyDeviceToDevice, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_graph::ensure_half() const {
    if (sym_state != 0) return SQGR_OK;
    sym_state = -1;
    if (nnz == 0 || nnz >= ((int64_t)1 << 32) || getenv("SQGR_NO_SYMMETRY")) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t ntiles = ceil_div(nnz, HALF_TILE);
    DevBuf<int> flags;
    DevBuf<uint32_t> t_lt, t_self;
    DevBuf<unsigned long long> totals;
    SQGR_TRY(flags.alloc(2));
    SQGR_HIP(hipMemsetAsync(flags.p, 0, 8, st));
    {
        LaunchTimer t(ctx, "graph_sym_check");
        k_sym_check<<<(unsigned)ceil_div(nnz, 256), 256, 0, st>>>(indptr.p, indices.p, erow.p, nnz, flags.p);
        SQGR_HIP(hipGetLastError());
    }
    int h_flags[2] = {1, 1};
    SQGR_HIP(hipMemcpyAsync(h_flags, flags.p, 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    canonical_rows = h_flags[1] == 0;
    if (h_flags[0] || h_flags[1]) return SQGR_OK;  // not symmetric (or not canonical): the full edge list is used
    SQGR_TRY(t_lt.alloc((size_t)ntiles));
    SQGR_TRY(t_self.alloc((size_t)ntiles));
    SQGR_TRY(totals.alloc(2));
    unsigned long long h_tot[2] = {0, 0};
    {
        LaunchTimer t(ctx, "graph_half_list");
        k_half_count<<<(unsigned)ntiles, HALF_TILE, 0, st>>>(coo.p, nnz, t_lt.p, t_self.p);
        k_half_scan<<<1, 1024, 0, st>>>(t_lt.p, t_self.p, ntiles, totals.p);
        SQGR_HIP(hipGetLastError());
        SQGR_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, st));
        SQGR_HIP(hipStreamSynchronize(st));
        if (2 * h_tot[0] + h_tot[1] != (unsigned long long)nnz) {
            set_error("internal error: half list of a symmetric graph has %llu + %llu entries for nnz=%lld", h_tot[0], h_tot[1],
                      (long long)nnz);
            return SQGR_ERR_HIP;
        }
        SQGR_TRY(half.alloc((size_t)(h_tot[0] + h_tot[1]) + LIST_PAD));
        SQGR_HIP(hipMemsetAsync(half.p + (h_tot[0] + h_tot[1]), 0, (size_t)LIST_PAD * sizeof(int2), st));
        k_half_scatter<<<(unsigned)ntiles, HALF_TILE, 0, st>>>(coo.p, nnz, t_lt.p, t_self.p, (uint32_t)h_tot[0], half.p);
        SQGR_HIP(hipGetLastError());
    }
    SQGR_HIP(hipStreamSynchronize(st));
    n_half = (int64_t)h_tot[0];
    n_self = (int64_t)h_tot[1];
    sym_state = 1;
    SQGR_TRY(tile_edge_list(ctx, half.p, n_half, n));
    return SQGR_OK;
}

int sqgr_graph::ensure_split() const {
    if (split_state != 0) return SQGR_OK;
    SQGR_TRY(ensure_half());
    split_state = -1;
    if (sym_state != -1 || !canonical_rows || nnz < 2 || nnz >= ((int64_t)1 << 31) || getenv("SQGR_NO_SPLIT")) return SQGR_OK;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t ntiles = ceil_div(nnz, HALF_TILE);
    DevBuf<int> flags;
    DevBuf<uint8_t> cls;
    DevBuf<uint32_t> t_a, t_b;
    DevBuf<unsigned long long> totals;
    SQGR_TRY(flags.alloc(2));
    SQGR_TRY(cls.alloc((size_t)nnz));
    SQGR_TRY(t_a.alloc((size_t)ntiles));
    SQGR_TRY(t_b.alloc((size_t)ntiles));
    SQGR_TRY(totals.alloc(2));
    SQGR_HIP(hipMemsetAsync(flags.p, 0, 8, st));
    LaunchTimer t(ctx, "graph_split_list");
    k_split_classify<<<(unsigned)ceil_div(nnz, 256), 256, 0, st>>>(indptr.p, indices.p, erow.p, nnz, cls.p, flags.p);
    k_half_count<<<(unsigned)ntiles, HALF_TILE, 0, st>>>(coo.p, nnz, t_a.p, t_b.p, cls.p);
    k_half_scan<<<1, 1024, 0, st>>>(t_a.p, t_b.p, ntiles, totals.p);
    SQGR_HIP(hipGetLastError());
    int h_flags[2] = {1, 1};
    unsigned long long h_tot[2] = {0, 0};
    SQGR_HIP(hipMemcpyAsync(h_flags, flags.p, 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    if (h_flags[0]) return SQGR_OK;                                                    // self loops
    if (2 * h_tot[0] + h_tot[1] != (unsigned long long)nnz) {
        set_error("internal error: split list of a directed graph has 2 x %llu + %llu entries for nnz=%lld", h_tot[0], h_tot[1], (long long)nnz);
        return SQGR_ERR_HIP;
    }
    if (h_tot[0] * 8 < (unsigned long long)nnz) return SQGR_OK;                        // fewer than an eighth of the entries spared
    const size_t m = (size_t)h_tot[0], o = (size_t)h_tot[1];
    SQGR_TRY(split.alloc(m + o + 2 * (size_t)LIST_PAD));
    SQGR_HIP(hipMemsetAsync(split.p + m, 0, (size_t)LIST_PAD * sizeof(int2), st));
    SQGR_HIP(hipMemsetAsync(split.p + m + LIST_PAD + o, 0, (size_t)LIST_PAD * sizeof(int2), st));
    k_half_scatter<<<(unsigned)ntiles, HALF_TILE, 0, st>>>(coo.p, nnz, t_a.p, t_b.p, (uint32_t)(m + LIST_PAD), split.p, cls.p);
    SQGR_HIP(hipGetLastError());
    SQGR_HIP(hipStreamSynchronize(st));
    n_mutual = (int64_t)m;
    n_oneway = (int64_t)o;
    split_state = 1;
    return SQGR_OK;
}

__global__ __launch_bounds__(256) void k_pass_list(const int2* __restrict__ src, uint32_t m, uint32_t total, uint32_t J, uint32_t R, int shift,
                                                   int2* __restrict__ dst) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= total) return;
    const uint32_t l = pass_list_logical(p, J, R);
    int2 v = l < m ? src[l] : make_int2(0, 0);
    v.x = (int)((uint32_t)v.x >> shift);  // 16 * spot -> w * spot
    v.y = (int)((uint32_t)v.y >> shift);
    dst[p] = v;
}

int sqgr_graph::pass_list(int J, int R, int w, const int2** out) const {
    SQGR_TRY(ensure_half());
    const bool is_half = sym_state == 1;
    const int2* src = is_half ? half.p : coo.p;
    const int64_t m = is_half ? n_half + n_self : nnz;
    SQGR_REQUIRE((J == 32 || J == 64) && (R == 1 || R == 2 || R == 4) && (w == 8 || w == 4 || w == 2 || w == 1), "pass list (%d, %d, %d)", J, R, w);
    const int shift = w == 8 ? 1 : w == 4 ? 2 : w == 2 ? 3 : 4;
    PassList& pl = pass_lists[shift - 1];
    if (pl.J != J || pl.R != R || pl.w != w || !pl.list.p) {
        SQGR_HIP(hipSetDevice(ctx->device));
        const int64_t G = 4 * (int64_t)J;
        const int64_t total = ceil_div(m, G) * G + LIST_PAD;
        pl.list.release();
        SQGR_TRY(pl.list.alloc((size_t)total));
        LaunchTimer t(ctx, "graph_pass_list");
        k_pass_list<<<(unsigned)ceil_div(total, 256), 256, 0, ctx->stream>>>(src, (uint32_t)m, (uint32_t)total, (uint32_t)J, (uint32_t)R, shift, pl.list.p);
        SQGR_HIP(hipGetLastError());
        pl.J = J;
        pl.R = R;
        pl.w = w;
    }
    *out = pl.list.p;
    return SQGR_OK;
}

// one block per group of G = 4*J physical entries: the group's smallest row is its base; flag[0] != 0: some entry does not fit
__global__ __launch_bounds__(256) void k_packed_list(const int2* __restrict__ src, uint32_t m, uint32_t J, uint32_t R,
                                                    uint32_t* __restrict__ dst, uint32_t* __restrict__ base, int* __restrict__ flag) {
    __shared__ uint32_t s_min;
    const uint32_t G = 4 * J, p0 = blockIdx.x * G;
    if (threadIdx.x == 0) s_min = 0xffffffffu;
    __syncthreads();
    int2 v = make_int2(0, 0);
    bool real = false;
    if (threadIdx.x < G) {
        const uint32_t l = pass_list_logical(p0 + threadIdx.x, J, R);
        real = l < m;
        if (real) {
            v = src[l];
            atomicMin(&s_min, (uint32_t)v.x >> 4);
        }
    }
    __syncthreads();
    const uint32_t b = s_min == 0xffffffffu ? 0u : s_min;
    if (threadIdx.x == 0) base[blockIdx.x] = b;
    if (threadIdx.x >= G) return;
    uint32_t e = 0;  // padding: the base row against itself (a valid spot; the kernel's tail logic adds 0 for it)
    if (real) {
        const uint32_t r = (uint32_t)v.x >> 4, c = (uint32_t)v.y >> 4;
        const int32_t d = (int32_t)c - (int32_t)r;
        if (r - b > 255u || d >= (1 << 23) || d < -(1 << 23)) atomicOr(flag, 1);
        e = ((uint32_t)d << 8) | ((r - b) & 255u);
    }
    dst[p0 + threadIdx.x] = e;
}

int sqgr_graph::packed_list(int J, int R, const uint32_t** out_list, const uint32_t** out_base) const {
    *out_list = *out_base = nullptr;
    SQGR_TRY(ensure_half());
    PackedList& pl = packed_lists[J == 32 ? 0 : 1];
    if (pl.state == 0 || pl.J != J || pl.R != R) {
        const bool is_half = sym_state == 1;
        const int2* src = is_half ? half.p : coo.p;
        const int64_t m = is_half ? n_half + n_self : nnz;
        SQGR_HIP(hipSetDevice(ctx->device));
        const int64_t G = 4 * (int64_t)J;
        const int64_t groups = ceil_div(m, G) + ceil_div((int64_t)LIST_PAD, G) + 1;
        pl.list.release();
        pl.base.release();
        SQGR_TRY(pl.list.alloc((size_t)(groups * G)));
        SQGR_TRY(pl.base.alloc((size_t)groups));
        DevBuf<int> flag;
        SQGR_TRY(flag.alloc(1));
        SQGR_HIP(hipMemsetAsync(flag.p, 0, 4, ctx->stream));
        {
            LaunchTimer t(ctx, "graph_packed_list");
            k_packed_list<<<(unsigned)groups, 256, 0, ctx->stream>>>(src, (uint32_t)m, (uint32_t)J, (uint32_t)R, pl.list.p, pl.base.p, flag.p);
            SQGR_HIP(hipGetLastError());
        }
        int h_flag = 1;
        SQGR_HIP(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        SQGR_HIP(hipStreamSynchronize(ctx->stream));
        pl.J = J;
        pl.R = R;
        pl.state = h_flag ? -1 : 1;
        if (h_flag) {
            pl.list.release();
            pl.base.release();
        }
    }
    if (pl.state == 1) {
        *out_list = pl.list.p;
        *out_base = pl.base.p;
    }
    return SQGR_OK;
}

int sqgr_ctx::timer_id(const char* name) {
    auto it = timer_ids.find(name);
    if (it != timer_ids.end()) return it->second;
    int id = (int)timer_names.size();
    timer_names.emplace_back(name);
    timer_ids[name] = id;
    timer_ms.push_back(0.0);
    timer_count.push_back(0);
    return id;
}

int sqgr_ctx::begin_launch(const char* name, TimedLaunch* tl, hipStream_t st) {
    tl->name_id = timer_id(name);
    tl->stream = st;
    for (hipEvent_t* ev : {&tl->start, &tl->stop}) {
        if (!event_pool.empty()) {
            *ev = event_pool.back();
            event_pool.pop_back();
        } else {
            SQGR_HIP(hipEventCreate(ev));
        }
    }
    SQGR_HIP(hipEventRecord(tl->start, st));
    return SQGR_OK;
}

int sqgr_ctx::end_launch(const TimedLaunch& tl) {
    SQGR_HIP(hipEventRecord(tl.stop, tl.stream));
    launches.push_back(tl);
    if (launches.size() >= 8192) return resolve_timers();
    return SQGR_OK;
}

int sqgr_ctx::resolve_timers() {
    if (launches.empty()) return SQGR_OK;
    SQGR_HIP(hipStreamSynchronize(stream));
    SQGR_HIP(hipStreamSynchronize(stream2));
    for (const TimedLaunch& tl : launches) {
        float ms = 0.f;
        SQGR_HIP(hipEventElapsedTime(&ms, tl.start, tl.stop));
        timer_ms[tl.name_id] += ms;
        timer_count[tl.name_id] += 1;
        event_pool.push_back(tl.start);
        event_pool.push_back(tl.stop);
    }
    launches.clear();
    return SQGR_OK;
}

extern "C" {

int sqgr_abi_version(void) { return SQGR_ABI_VERSION; }

const char* sqgr_last_error(void) { return g_err; }

int sqgr_device_count(int* out_count) {
    SQGR_REQUIRE(out_count, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return SQGR_ERR_NODEVICE;
    }
    *out_count = n;
    return SQGR_OK;
}

int sqgr_ctx_create(int device, sqgr_ctx** out_ctx) {
    SQGR_REQUIRE(out_ctx, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (libsqgr has no CPU fallback)");
        return SQGR_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range [0,%d)", device, n);
        return SQGR_ERR_NODEVICE;
    }
    SQGR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SQGR_HIP(hipGetDeviceProperties(&prop, device));
    sqgr_ctx* ctx = new sqgr_ctx();
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return SQGR_ERR_HIP;
    }
    sqgr::pool_register_streams(device, ctx->stream, ctx->stream2);
    *out_ctx = ctx;
    return SQGR_OK;
}

int sqgr_ctx::scratch_get(int slot, size_t bytes, void** out) {
    if ((size_t)slot >= scratch.size()) scratch.resize((size_t)slot + 1, {nullptr, 0});
    auto& sc = scratch[(size_t)slot];
    if (bytes == 0) bytes = 8;
    if (sc.second < bytes) {
        if (sc.first) (void)sqgr::dev_free(sc.first);
        sc = {nullptr, 0};
        const size_t want = bytes + bytes / 4;  // head-room: point counts of successive calls vary a little
        hipError_t e = sqgr::dev_malloc(&sc.first, want);
        if (e != hipSuccess) {
            sc.first = nullptr;
            sqgr::set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            return SQGR_ERR_NOMEM;
        }
        sc.second = want;
    }
    *out = sc.first;
    return SQGR_OK;
}

int sqgr_ctx_destroy(sqgr_ctx* ctx) {
    if (!ctx) return SQGR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    sqgr::pool_unregister_streams(ctx->stream, ctx->stream2);
    sqgr::flush_deferred_frees();
    sqgr::pool_flush();
    for (auto& tl : ctx->launches) {
        (void)hipEventDestroy(tl.start);
        (void)hipEventDestroy(tl.stop);
    }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    delete ctx->autocorr_lists;
    ctx->autocorr_lists = nullptr;
    for (auto& sc : ctx->scratch)
        if (sc.first) (void)sqgr::dev_free(sc.first);
    (void)hipStreamDestroy(ctx->stream);
    (void)hipStreamDestroy(ctx->stream2);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
    return SQGR_OK;
}

int sqgr_ctx_trim(sqgr_ctx* ctx, int64_t keep_bytes) {
    SQGR_REQUIRE(ctx && keep_bytes >= 0, "ctx is NULL or keep_bytes < 0");
    SQGR_HIP(hipSetDevice(ctx->device));
    SQGR_HIP(hipDeviceSynchronize());  // nothing of a previous owner may be in flight when a parked buffer goes back to the driver
    sqgr::flush_deferred_frees();
    sqgr::pool_trim(ctx->device, (size_t)keep_bytes);
    return SQGR_OK;
}

int sqgr_debug_counters(int64_t* out, int32_t count) {
    SQGR_REQUIRE(out && count >= 0, "out is NULL or count < 0");
    const sqgr::AllocStats& a = sqgr::g_alloc_stats;
    const int64_t v[8] = {a.mallocs.load(), a.malloc_bytes.load(), a.malloc_ns.load(), a.frees.load(),
                          a.free_ns.load(), a.pool_hits.load(), a.pool_parks.load(), a.pool_flushes.load()};
    for (int i = 0; i < count; ++i) out[i] = i < 8 ? v[i] : 0;
    return SQGR_OK;
}

int sqgr_ctx_sync(sqgr_ctx* ctx) {
    SQGR_REQUIRE(ctx, "ctx is NULL");
    SQGR_HIP(hipSetDevice(ctx->device));
    SQGR_HIP(hipStreamSynchronize(ctx->stream));
    SQGR_HIP(hipStreamSynchronize(ctx->stream2));
    return SQGR_OK;
}

int sqgr_ctx_device_info(sqgr_ctx* ctx, char* name, int len, int* cu_count, int64_t* hbm_bytes) {
    SQGR_REQUIRE(ctx, "ctx is NULL");
    hipDeviceProp_t prop;
    SQGR_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && len > 0) {
        snprintf(name, (size_t)len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SQGR_OK;
}

int sqgr_timer_enable(sqgr_ctx* ctx, int enable) {
    SQGR_REQUIRE(ctx, "ctx is NULL");
    if (!enable) SQGR_TRY(ctx->resolve_timers());
    ctx->timing = enable != 0;
    return SQGR_OK;
}

int sqgr_timer_reset(sqgr_ctx* ctx) {
    SQGR_REQUIRE(ctx, "ctx is NULL");
    SQGR_TRY(ctx->resolve_timers());
    for (auto& v : ctx->timer_ms) v = 0.0;
    for (auto& v : ctx->timer_count) v = 0;
    return SQGR_OK;
}

int sqgr_timer_get(sqgr_ctx* ctx, const char* prefix, double* total_ms, int64_t* launches) {
    SQGR_REQUIRE(ctx && prefix, "ctx/prefix is NULL");
    SQGR_TRY(ctx->resolve_timers());
    double ms = 0.0;
    int64_t cnt = 0;
    size_t plen = strlen(prefix);
    for (size_t i = 0; i < ctx->timer_names.size(); ++i) {
        if (ctx->timer_names[i].compare(0, plen, prefix) == 0) {
            ms += ctx->timer_ms[i];
            cnt += ctx->timer_count[i];
        }
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    return SQGR_OK;
}

int sqgr_timer_report(sqgr_ctx* ctx, char* buf, int len) {
    SQGR_REQUIRE(ctx && buf && len > 0, "ctx/buf is NULL");
    SQGR_TRY(ctx->resolve_timers());
    std::string s;
    for (size_t i = 0; i < ctx->timer_names.size(); ++i) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s:%lld:%.6f;", ctx->timer_names[i].c_str(), (long long)ctx->timer_count[i],
                 ctx->timer_ms[i]);
        s += tmp;
    }
    snprintf(buf, (size_t)len, "%s", s.c_str());
    return SQGR_OK;
}

static int graph_create_impl(sqgr_ctx* ctx, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                             const float* data32, const double* data, sqgr_graph** out_graph) {
    SQGR_REQUIRE(ctx && out_graph, "ctx/out_graph is NULL");
    *out_graph = nullptr;
    SQGR_REQUIRE(n > 0 && n < (int64_t)0x7fffffff, "n=%lld out of range", (long long)n);
    SQGR_REQUIRE(nnz >= 0, "nnz=%lld negative", (long long)nnz);
    SQGR_REQUIRE(indptr && (indices || nnz == 0), "indptr/indices is NULL");
    SQGR_REQUIRE(indptr[0] == 0 && indptr[n] == nnz, "indptr[0]=%lld indptr[n]=%lld inconsistent with nnz=%lld",
                 (long long)indptr[0], (long long)indptr[n], (long long)nnz);
    int64_t max_row_len = 0;
    for (int64_t i = 0; i < n; ++i) {
        SQGR_REQUIRE(indptr[i] <= indptr[i + 1], "indptr not monotone at row %lld", (long long)i);
        max_row_len = std::max(max_row_len, indptr[i + 1] - indptr[i]);
    }
    for (int64_t e = 0; e < nnz; ++e)
        SQGR_REQUIRE(indices[e] >= 0 && indices[e] < n, "indices[%lld]=%d out of [0,%lld)", (long long)e, indices[e],
                     (long long)n);
    std::vector<double> widened;
    if (data32) {  // float32 weights are widened exactly (after the arguments have been checked); every kernel computes with
        widened.assign(data32, data32 + nnz);  // float64 weights like the reference
        data = widened.data();
    }
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_graph* g = new sqgr_graph();
    g->ctx = ctx;
    g->n = n;
    g->nnz = nnz;
    g->max_row_len = max_row_len;
    int rc = SQGR_OK;
    do {
        if ((rc = g->indptr.alloc((size_t)n + 1)) != SQGR_OK) break;
        if ((rc = g->indices.alloc((size_t)nnz)) != SQGR_OK) break;
        if ((rc = g->erow.alloc((size_t)nnz)) != SQGR_OK) break;
        if ((rc = g->coo.alloc((size_t)nnz + LIST_PAD)) != SQGR_OK) break;
        if (hipMemsetAsync(g->coo.p + nnz, 0, (size_t)LIST_PAD * sizeof(int2), ctx->stream) != hipSuccess) { rc = SQGR_ERR_HIP; break; }
        if (data && (rc = g->data.alloc((size_t)nnz)) != SQGR_OK) break;
        hipError_t e = hipMemcpyAsync(g->indptr.p, indptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nnz)
            e = hipMemcpyAsync(g->indices.p, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && data && nnz)
            e = hipMemcpyAsync(g->data.p, data, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nnz) {
            LaunchTimer t(ctx, "graph_expand_rows");
            k_expand_rows<<<(unsigned)ceil_div(nnz, 256), 256, 0, ctx->stream>>>(g->indptr.p, g->indices.p, n, nnz, g->erow.p, g->coo.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("graph upload failed: %s", hipGetErrorString(e));
            rc = SQGR_ERR_HIP;
        }
    } while (0);
    if (rc != SQGR_OK) {
        delete g;
        return rc;
    }
    g->has_data = data != nullptr;
    *out_graph = g;
    return SQGR_OK;
}

int sqgr_graph_create(sqgr_ctx* ctx, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                      const float* data, sqgr_graph** out_graph) {
    return graph_create_impl(ctx, n, nnz, indptr, indices, data, nullptr, out_graph);
}

int sqgr_graph_create_f64(sqgr_ctx* ctx, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                          const double* data, sqgr_graph** out_graph) {
    return graph_create_impl(ctx, n, nnz, indptr, indices, nullptr, data, out_graph);
}

int sqgr_graph_renumbered(sqgr_ctx* ctx, const sqgr_graph* g, const int32_t* order, sqgr_graph** out_graph) {
    SQGR_REQUIRE(ctx && g && order && out_graph, "null argument");
    *out_graph = nullptr;
    SQGR_REQUIRE(g->ctx == ctx, "graph belongs to a different context");
    SQGR_REQUIRE(g->max_row_len <= 4096, "a row of %lld entries: rows are sorted by insertion", (long long)g->max_row_len);
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = g->n, nnz = g->nnz;
    DevBuf<int32_t> d_order, d_pos;
    DevBuf<int64_t> deg;
    DevBuf<int> flags;
    SQGR_TRY(d_order.alloc((size_t)n));
    SQGR_TRY(d_pos.alloc((size_t)n));
    SQGR_TRY(deg.alloc((size_t)n + 1));
    SQGR_TRY(flags.alloc(2));
    sqgr_graph* t = new sqgr_graph();
    t->ctx = ctx;
    t->n = n;
    t->nnz = nnz;
    t->max_row_len = g->max_row_len;
    auto fail = [&](int code) {
        delete t;
        return code;
    };
    int rc = SQGR_OK;
    if ((rc = t->indptr.alloc((size_t)n + 1)) || (rc = t->indices.alloc((size_t)std::max<int64_t>(nnz, 1))) ||
        (rc = t->erow.alloc((size_t)std::max<int64_t>(nnz, 1))) || (rc = t->coo.alloc((size_t)nnz + LIST_PAD)))
        return fail(rc);
    hipError_t e = hipMemcpyAsync(d_order.p, order, (size_t)n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_pos.p, 0xFF, (size_t)n * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(flags.p, 0, 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(t->coo.p + nnz, 0, (size_t)LIST_PAD * sizeof(int2), st);
    if (e == hipSuccess) e = hipMemsetAsync(deg.p + n, 0, 8, st);
    if (e != hipSuccess) {
        set_error("graph renumbering failed: %s", hipGetErrorString(e));
        return fail(SQGR_ERR_HIP);
    }
    {
        LaunchTimer tm(ctx, "graph_renumber");
        const unsigned gn = (unsigned)ceil_div(n, 256);
        k_order_inverse<<<gn, 256, 0, st>>>(d_order.p, n, d_pos.p, flags.p);
        k_order_degrees<<<gn, 256, 0, st>>>(g->indptr.p, d_order.p, n, deg.p);
        size_t tmp_bytes = 0;
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, deg.p, t->indptr.p, (int)(n + 1), st);
        DevBuf<uint8_t> tmp;
        if (e == hipSuccess && (rc = tmp.alloc(tmp_bytes)) != SQGR_OK) return fail(rc);
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, deg.p, t->indptr.p, (int)(n + 1), st);
        int h_flags[2] = {1, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags.p, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess && h_flags[0]) {
            set_error("`order` is not a permutation of 0 .. %lld", (long long)(n - 1));
            return fail(SQGR_ERR_INVALID);
        }
        if (e == hipSuccess && nnz) {
            k_order_rows<<<gn, 256, 0, st>>>(g->indptr.p, g->indices.p, d_order.p, d_pos.p, t->indptr.p, n, t->indices.p);
            k_expand_rows<<<(unsigned)ceil_div(nnz, 256), 256, 0, st>>>(t->indptr.p, t->indices.p, n, nnz, t->erow.p, t->coo.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        set_error("graph renumbering failed: %s", hipGetErrorString(e));
        return fail(SQGR_ERR_HIP);
    }
    t->has_data = false;  // (structure only: what the permutation test reads)
    *out_graph = t;
    return SQGR_OK;
}

int sqgr_spatial_order(sqgr_ctx* ctx, const double* xy, int64_t n, int32_t* out_order) {
    SQGR_REQUIRE(ctx && xy && out_order && n > 0 && n < (int64_t)0x7fffffff, "null argument or n out of range");
    double lo[2] = {xy[0], xy[1]}, hi[2] = {xy[0], xy[1]};
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < 2; ++d) {
            const double v = xy[2 * i + d];
            SQGR_REQUIRE(v == v && v - v == 0.0, "coordinate %lld is not finite", (long long)i);
            lo[d] = std::min(lo[d], v);
            hi[d] = std::max(hi[d], v);
        }
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<double> d_xy;
    DevBuf<uint32_t> keys, keys_out;
    DevBuf<int32_t> idx, idx_out;
    SQGR_TRY(d_xy.alloc((size_t)2 * n));
    SQGR_TRY(keys.alloc((size_t)n));
    SQGR_TRY(keys_out.alloc((size_t)n));
    SQGR_TRY(idx.alloc((size_t)n));
    SQGR_TRY(idx_out.alloc((size_t)n));
    SQGR_HIP(hipMemcpyAsync(d_xy.p, xy, (size_t)n * 16, hipMemcpyHostToDevice, st));
    LaunchTimer tm(ctx, "graph_spatial_order");
    const double sx = hi[0] > lo[0] ? 65535.0 / (hi[0] - lo[0]) : 0.0, sy = hi[1] > lo[1] ? 65535.0 / (hi[1] - lo[1]) : 0.0;
    k_morton_keys<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(d_xy.p, n, lo[0], lo[1], sx, sy, keys.p, idx.p);
    SQGR_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    SQGR_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_out.p, idx.p, idx_out.p, (int)n, 0, 32, st));
    DevBuf<uint8_t> tmp;
    SQGR_TRY(tmp.alloc(tmp_bytes));
    SQGR_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys_out.p, idx.p, idx_out.p, (int)n, 0, 32, st));
    SQGR_HIP(hipMemcpyAsync(out_order, idx_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_graph_destroy(sqgr_graph* g) {
    if (!g) return SQGR_OK;
    (void)hipSetDevice(g->ctx->device);
    delete g;
    return SQGR_OK;
}

}  // extern "C"
