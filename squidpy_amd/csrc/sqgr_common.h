// Internal helpers shared by the libsqgr translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sqgr.h"

namespace sqgr {

void set_error(const char* fmt, ...);

#define SQGR_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e__ = (call);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            ::sqgr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (e__ == hipErrorOutOfMemory) ? SQGR_ERR_NOMEM : SQGR_ERR_HIP;                   \
        }                                                                                           \
    } while (0)

#define SQGR_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::sqgr::set_error(__VA_ARGS__); \
            return SQGR_ERR_INVALID;       \
        }                                  \
    } while (0)

#define SQGR_TRY(expr)            \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != SQGR_OK) return rc__; \
    } while (0)

struct TimedLaunch {
    int name_id;
    hipEvent_t start, stop;
    hipStream_t stream;
};

}  // namespace sqgr

namespace sqgr {
struct CtxCache {  // device-resident state a translation unit keeps in the context between calls (freed with the context)
    virtual ~CtxCache() {}
};
}  // namespace sqgr

struct sqgr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // producer stream of two-stage pipelines (label shuffles overlap counting)
    hipStream_t copy_stream = nullptr;  // host-to-device uploads that overlap the compute stream (sqgr_matrix_upload_columns: its own host thread)
    int cu_count = 0;
    bool timing = false;
    std::vector<std::string> timer_names;
    std::map<std::string, int> timer_ids;
    std::vector<sqgr::TimedLaunch> launches;      // unresolved event pairs
    std::vector<double> timer_ms;                 // resolved totals per name id
    std::vector<int64_t> timer_count;
    std::vector<hipEvent_t> event_pool;

    // small reusable device buffers for entry points that are called in tight host loops (Ripley: one call per cluster
    // and per simulation): hipMalloc/hipFree per call cost more than the kernels.  Slot s holds at least the bytes last
    // asked for; valid until the next request for the same slot.  Single stream, calls are synchronous: no aliasing.
    std::vector<std::pair<void*, size_t>> scratch;
    int scratch_get(int slot, size_t bytes, void** out);
    sqgr::CtxCache* autocorr_lists = nullptr;  // bucket lists of the last permutation set (sqgr_autocorr.hip)

    int timer_id(const char* name);
    int begin_launch(const char* name, sqgr::TimedLaunch* tl, hipStream_t st);
    int end_launch(const sqgr::TimedLaunch& tl);
    int resolve_timers();
};

namespace sqgr {

// roctx ranges (SQGR_ROCTX=1; libroctx64 bound at run time): every bracketed launch shows up under its name in a rocprofv3
// `--marker-trace` timeline — the tracing hook SURVEY §5 lists next to the HIP-event timers.  No-ops otherwise.
void roctx_push(const char* name);
void roctx_pop();

// RAII bracket: records start/stop events on ctx->stream around a kernel launch when timing is on.
struct LaunchTimer {
    sqgr_ctx* ctx;
    TimedLaunch tl;
    bool active;
    LaunchTimer(sqgr_ctx* c, const char* name, hipStream_t st = nullptr) : ctx(c), active(false) {
        roctx_push(name);
        if (ctx->timing) active = (ctx->begin_launch(name, &tl, st ? st : c->stream) == SQGR_OK);
    }
    ~LaunchTimer() {
        if (active) ctx->end_launch(tl);
        roctx_pop();
    }
};

// Parked device buffers (sqgr_ctx.hip).  One spatial_autocorr call over 20 000 genes allocates and frees ~100 GB (the 16 GB
// expression matrix, ~9 GB of working buffers per 2048-gene block); after two or three calls the driver had handed out all of
// the 288 GB once and the next hipMalloc stalled for ~5 s while freed memory was reclaimed (HIP API trace of bench.py: one
// hipMalloc of 1.6 GB taking 4.85 s).  So buffers of 64 MB and more are parked per device when their owner lets go of them and
// handed to the next request of about that size instead of going back to the driver.  A reused buffer is zeroed (a fresh
// allocation reads as zeros too) after a device synchronise (hipFree used to provide that one); alloc_pooled skips the zeroing
// for buffers whose every byte the owner writes.  Bounded (SQGR_POOL_GB, default 32; 0 switches it off; sqgr_ctx_trim hands parked
// buffers back on request — other HIP users of the process see parked memory as taken); flushed when any
// hipMalloc of the library fails (which is then retried) and when a context is destroyed.
constexpr size_t POOL_MIN_BYTES = (size_t)64 << 20;
// Every device allocation of the library goes through dev_malloc / dev_free (sqgr_ctx.hip): counted and timed, so that a caller
// — bench.py's legs, the "a second call allocates nothing" tests — can see what a call asked of the driver (sqgr_debug_counters).
struct AllocStats {
    std::atomic<int64_t> mallocs{0}, malloc_bytes{0}, malloc_ns{0}, frees{0}, free_ns{0}, pool_hits{0}, pool_parks{0}, pool_flushes{0};
};
extern AllocStats g_alloc_stats;
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
void streaming_upload_begin();  // between these two, dev_free puts its pointers aside (hipFree would wait for the upload's copy stream)
void streaming_upload_end();
void flush_deferred_frees();
void* pool_take(size_t bytes, size_t* capacity);  // a parked buffer of [bytes, 1.5 * bytes] on the current device, or NULL
void pool_give(void* p, size_t capacity);         // parks p or frees it
void pool_flush();                                // frees everything parked on the current device
void pool_trim(int device, size_t keep_bytes);    // frees parked buffers of `device` until at most keep_bytes stay parked
// Waits for the COMPUTE streams (stream, stream2) of every context on the current device — what a previous owner of a parked buffer
// can have in flight.  Not hipDeviceSynchronize: that also waits for the copy streams, i.e. for a whole matrix upload running on
// another host thread (sqgr_matrix_upload_columns), which then never overlaps the kernels it is meant to hide behind (round 6:
// the first feature block of config 3 became ready after all ten column blocks had arrived).
hipError_t pool_quiesce();
void pool_register_streams(int device, hipStream_t a, hipStream_t b);
void pool_unregister_streams(hipStream_t a, hipStream_t b);

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t pooled_bytes = 0;  // != 0: capacity of a buffer that goes back to the pool
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && pooled_bytes) pool_give(p, pooled_bytes);
        else if (p) (void)dev_free(p);
        p = nullptr;
        n = 0;
        pooled_bytes = 0;
    }
    int alloc_impl(size_t count, bool zero_reused) {
        release();
        if (count == 0) count = 1;
        const size_t bytes = count * sizeof(T);
        if (bytes >= POOL_MIN_BYTES) {
            size_t cap = 0;
            if (void* q = pool_take(bytes, &cap)) {
                hipError_t e = pool_quiesce();  // nothing of the previous owner is in flight any more
                if (e == hipSuccess && zero_reused) {
                    e = hipMemsetAsync(q, 0, bytes, nullptr);
                    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
                }
                if (e != hipSuccess) {
                    (void)dev_free(q);
                    set_error("reusing a parked buffer failed: %s", hipGetErrorString(e));
                    return SQGR_ERR_HIP;
                }
                p = static_cast<T*>(q);
                n = count;
                pooled_bytes = cap;
                return SQGR_OK;
            }
        }
        hipError_t e = dev_malloc(reinterpret_cast<void**>(&p), bytes);
        if (e == hipErrorOutOfMemory) {  // give the parked buffers (and the frees put aside during an upload) back to the driver and try once more
            (void)hipGetLastError();
            flush_deferred_frees();
            pool_flush();
            e = dev_malloc(reinterpret_cast<void**>(&p), bytes);
        }
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return SQGR_ERR_NOMEM;
        }
        n = count;
        pooled_bytes = bytes >= POOL_MIN_BYTES ? bytes : 0;
        return SQGR_OK;
    }
    int alloc(size_t count) { return alloc_impl(count, true); }
    int alloc_pooled(size_t count) { return alloc_impl(count, false); }  // contents undefined: the owner overwrites every byte
    int ensure(size_t count) { return (count <= n && p) ? SQGR_OK : alloc(count); }
    size_t bytes() const { return n * sizeof(T); }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// zero entries behind the nhood edge lists (sqgr_graph::coo / half): the count kernels' look-ahead loads stay in bounds.  A block's
// sweep runs ceil((T + 2) / 3) * 3 stages and stage t loads the entries of iteration t + 2: the last load ends at e0 + (T + 6) * STEP
// with e0 + T * STEP < m + STEP for the tail chunk — SEVEN iterations past the list's end for the widest kernel (one lane per edge,
// STEP = 4096 edges per iteration of a block), plus a lane's own four edges.  (Round 5 padded six: ADVICE r5, a third of all graph
// sizes read up to 30 KB past the allocation — values never consumed, but an out-of-bounds read all the same.)
constexpr int LIST_PAD = 7 * 4096 + 64;

// device-side collectives of an sqgr_comm (sqgr_comm.hip); no-ops for a NULL communicator or a single rank
int comm_allreduce_i64_dev(sqgr_comm* c, int64_t* dev_buf, size_t count, bool op_max, hipStream_t st);
int comm_allgather_dev(sqgr_comm* c, const void* dev_send, void* dev_recv, size_t bytes_per_rank, hipStream_t st);
int comm_agree(sqgr_comm* c, int local_rc, hipStream_t st);  // error agreement in front of a data collective (sqgr_comm.hip)
int comm_rank(const sqgr_comm* c);
int comm_world(const sqgr_comm* c);

}  // namespace sqgr

struct sqgr_graph {
    sqgr_ctx* ctx = nullptr;
    int64_t n = 0, nnz = 0;
    int64_t max_row_len = 0;        // most stored entries of one row (the 16-bit count kernels bound a counter with it)
    sqgr::DevBuf<int64_t> indptr;   // [n+1]
    sqgr::DevBuf<int32_t> indices;  // [nnz]
    sqgr::DevBuf<int32_t> erow;     // [nnz] row of every stored edge (COO expansion, built on device)
    sqgr::DevBuf<int2> coo;         // [nnz + LIST_PAD] (16*row, 16*col): byte offsets of the endpoints' 16-byte label rows,
                                    // what the nhood count kernel gathers with; zero padding behind the list
    sqgr::DevBuf<double> data;      // [nnz] edge weights in float64 (float32 input is widened exactly) or empty
    bool has_data = false;
    // Structurally symmetric graphs (every stored (r, c) has a stored (c, r); canonical CSR: rows sorted, no duplicates):
    // the neighbourhood counts of a labelling satisfy count = h + h^T with h taken over the edges r < c only, so the
    // permutation kernels walk `half` = [edges with r < c in CSR order | self loops] — half the gathers and LDS atomics.
    // Built lazily on the device by ensure_half() (sqgr_ctx.hip); sym_state: 0 unknown, 1 symmetric, -1 not.
    mutable int sym_state = 0;
    mutable sqgr::DevBuf<int2> half;
    mutable int64_t n_half = 0;  // edges with r < c
    mutable int64_t n_self = 0;  // self loops, stored behind them
    int ensure_half() const;
    // Directed graphs (kNN graphs are, gr/neighbors.py: KNNBuilder does not symmetrise): most of their edges still come in mutual
    // pairs, and a pair needs ONE walk — its contribution to the counts is h + h^T like a half edge's.  `split` =
    // [mutual edges with r < c | LIST_PAD zeros | edges without a mirror | LIST_PAD zeros], built on first use on graphs in
    // canonical CSR form without self loops; split_state: 0 not tried, 1 built, -1 not applicable (symmetric, self loops, not
    // canonical, or fewer than an eighth of the entries would be spared).
    mutable int split_state = 0;
    mutable bool canonical_rows = false;  // ensure_half's check: rows strictly increasing
    mutable sqgr::DevBuf<int2> split;
    mutable int64_t n_mutual = 0;  // mutual pairs (edges with r < c that have a mirror)
    mutable int64_t n_oneway = 0;  // edges without a mirror
    int ensure_split() const;
    // The pass kernel of the permutation test (51 <= K <= 202 clusters, sqgr_nhood.hip: k_count_pass) reads the SAME list in
    // another order and with other offsets: inside every aligned group of 4*J entries (J = 32 | 64 edge slots of a wavefront)
    // slot j of gather instruction u = 0..3 — physical position 4*j + u — holds the list's entry (u/R)*(J*R) + j*R + u%R, so
    // that with R = 1 one gather instruction touches the label rows of J consecutive edges instead of every fourth edge of the
    // group (R = 4: the list's own order); and the entries are byte offsets (w*row, w*col) into a PLANE of w = 8 | 4 | 2 | 1
    // label bytes per spot instead of (16*row, 16*col).  pass_list builds the copy on first use (half list on symmetric
    // graphs, the full list otherwise; length rounded up to whole groups + LIST_PAD zero entries).
    struct PassList {
        int J = 0, R = 0, w = 0;
        sqgr::DevBuf<int2> list;
    };
    mutable PassList pass_lists[4];
    int pass_list(int J, int R, int w, const int2** out) const;
    // The same lists in 4 bytes per entry (round 5: the pass kernel is bound by the bytes it pulls from L2 into L1, and at 8, 4,
    // 2, 1 permutations per pass the 8-byte list entries are most of them): entry = (col - row) << 8 | (row - base of its group),
    // one base spot per group of 4*J entries in `base` — spot indices, scaled by the plane width in the kernel.  Admissible when
    // every group spans fewer than 256 rows and |col - row| < 2^23; state: 0 not tried, 1 built, -1 this graph does not fit (the
    // 8-byte list is used).  One per J (the order R is fixed when it is built).
    struct PackedList {
        int J = 0, R = 0, state = 0;
        sqgr::DevBuf<uint32_t> list, base;
    };
    mutable PackedList packed_lists[2];
    int packed_list(int J, int R, const uint32_t** out_list, const uint32_t** out_base) const;  // *out_list == nullptr: not admissible
};

namespace sqgr {
// logical position (in the list's own order) of physical position p of a pass list
__host__ __device__ inline uint32_t pass_list_logical(uint32_t p, uint32_t J, uint32_t R) {
    const uint32_t G = 4 * J, base = p / G * G, w = p - base, j = w >> 2, u = w & 3;
    return base + (u / R) * (J * R) + j * R + (u % R);
}
}  // namespace sqgr
