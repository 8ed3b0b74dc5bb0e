// libsqgr: context, error reporting, kernel timers and the device-resident CSR graph.
#include "sqgr_common.h"

namespace sqgr {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
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
    coo[e] = make_int2((int)lo, indices[e]);
}

}  // namespace sqgr

using namespace sqgr;

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
    if (e != hipSuccess) {
        delete ctx;
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return SQGR_ERR_HIP;
    }
    *out_ctx = ctx;
    return SQGR_OK;
}

int sqgr_ctx::scratch_get(int slot, size_t bytes, void** out) {
    if ((size_t)slot >= scratch.size()) scratch.resize((size_t)slot + 1, {nullptr, 0});
    auto& sc = scratch[(size_t)slot];
    if (bytes == 0) bytes = 8;
    if (sc.second < bytes) {
        if (sc.first) (void)hipFree(sc.first);
        sc = {nullptr, 0};
        const size_t want = bytes + bytes / 4;  // head-room: point counts of successive calls vary a little
        hipError_t e = hipMalloc(&sc.first, want);
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
    for (auto& tl : ctx->launches) {
        (void)hipEventDestroy(tl.start);
        (void)hipEventDestroy(tl.stop);
    }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    for (auto& sc : ctx->scratch)
        if (sc.first) (void)hipFree(sc.first);
    (void)hipStreamDestroy(ctx->stream);
    (void)hipStreamDestroy(ctx->stream2);
    delete ctx;
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

int sqgr_graph_create(sqgr_ctx* ctx, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                      const float* data, sqgr_graph** out_graph) {
    SQGR_REQUIRE(ctx && out_graph, "ctx/out_graph is NULL");
    *out_graph = nullptr;
    SQGR_REQUIRE(n > 0 && n < (int64_t)0x7fffffff, "n=%lld out of range", (long long)n);
    SQGR_REQUIRE(nnz >= 0, "nnz=%lld negative", (long long)nnz);
    SQGR_REQUIRE(indptr && (indices || nnz == 0), "indptr/indices is NULL");
    SQGR_REQUIRE(indptr[0] == 0 && indptr[n] == nnz, "indptr[0]=%lld indptr[n]=%lld inconsistent with nnz=%lld",
                 (long long)indptr[0], (long long)indptr[n], (long long)nnz);
    for (int64_t i = 0; i < n; ++i)
        SQGR_REQUIRE(indptr[i] <= indptr[i + 1], "indptr not monotone at row %lld", (long long)i);
    for (int64_t e = 0; e < nnz; ++e)
        SQGR_REQUIRE(indices[e] >= 0 && indices[e] < n, "indices[%lld]=%d out of [0,%lld)", (long long)e, indices[e],
                     (long long)n);
    SQGR_HIP(hipSetDevice(ctx->device));
    sqgr_graph* g = new sqgr_graph();
    g->ctx = ctx;
    g->n = n;
    g->nnz = nnz;
    int rc = SQGR_OK;
    do {
        if ((rc = g->indptr.alloc((size_t)n + 1)) != SQGR_OK) break;
        if ((rc = g->indices.alloc((size_t)nnz)) != SQGR_OK) break;
        if ((rc = g->erow.alloc((size_t)nnz)) != SQGR_OK) break;
        if ((rc = g->coo.alloc((size_t)nnz)) != SQGR_OK) break;
        if (data && (rc = g->data.alloc((size_t)nnz)) != SQGR_OK) break;
        hipError_t e = hipMemcpyAsync(g->indptr.p, indptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nnz)
            e = hipMemcpyAsync(g->indices.p, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && data && nnz)
            e = hipMemcpyAsync(g->data.p, data, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
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

int sqgr_graph_destroy(sqgr_graph* g) {
    if (!g) return SQGR_OK;
    (void)hipSetDevice(g->ctx->device);
    delete g;
    return SQGR_OK;
}

}  // extern "C"
