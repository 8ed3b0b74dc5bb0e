// libsqgr: the path's one real exchange — an all-reduce of exact integer accumulators over RCCL (xGMI) — behind the
// C ABI.  One process per GPU; the caller moves the 128-byte ncclUniqueId of rank 0 to the other
// ranks by whatever side channel it has (squidpy_amd/_dist.py: a socket rendezvous, or an existing torch.distributed
// group) and every rank calls sqgr_comm_create.
//
// RCCL is bound at run time (dlopen of librccl.so.1, resolved through this library's RUNPATH = the ROCm that libsqgr was
// built against, so that the communicator and libsqgr share ONE HIP runtime): single-GPU installations never need it,
// and the symbols cannot be captured by a second RCCL copy that some other package (PyTorch bundles its own) may have
// loaded into the process with RTLD_GLOBAL.
#include "sqgr_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace sqgr {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load() {
    if (g_rccl.handle) return SQGR_OK;
    const char* override_path = getenv("SQGR_RCCL_PATH");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        set_error("RCCL is not available: dlopen(librccl.so.1) failed: %s", dlerror());
        return SQGR_ERR_UNSUPPORTED;
    }
    RcclApi api;
    api.handle = h;
#define SQGR_SYM(field, name)                                                              \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));                     \
    if (!api.field) {                                                                      \
        set_error("RCCL symbol %s not found: %s", name, dlerror());                        \
        dlclose(h);                                                                        \
        return SQGR_ERR_UNSUPPORTED;                                                       \
    }
    SQGR_SYM(GetUniqueId, "ncclGetUniqueId")
    SQGR_SYM(CommInitRank, "ncclCommInitRank")
    SQGR_SYM(CommDestroy, "ncclCommDestroy")
    SQGR_SYM(AllReduce, "ncclAllReduce")
    SQGR_SYM(AllGather, "ncclAllGather")
    SQGR_SYM(Broadcast, "ncclBroadcast")
    SQGR_SYM(CommCount, "ncclCommCount")
    SQGR_SYM(CommUserRank, "ncclCommUserRank")
    SQGR_SYM(GetErrorString, "ncclGetErrorString")
#undef SQGR_SYM
    g_rccl = api;
    return SQGR_OK;
}

#define SQGR_NCCL(call)                                                                                   \
    do {                                                                                                  \
        ncclResult_t r__ = (call);                                                                        \
        if (r__ != ncclSuccess) {                                                                         \
            ::sqgr::set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
            return SQGR_ERR_HIP;                                                                          \
        }                                                                                                 \
    } while (0)

}  // namespace sqgr

using namespace sqgr;

struct sqgr_comm {
    sqgr_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf<int64_t> stage;  // staging buffer of the host-pointer entry points
};

// device-side collectives used by the other translation units (declared in sqgr_common.h)
namespace sqgr {

int comm_allreduce_i64_dev(sqgr_comm* c, int64_t* dev_buf, size_t count, bool op_max, hipStream_t st) {
    if (!c || c->world <= 1 || count == 0) return SQGR_OK;
    LaunchTimer t(c->ctx, "rccl_allreduce_i64", st);
    SQGR_NCCL(g_rccl.AllReduce(dev_buf, dev_buf, count, ncclInt64, op_max ? ncclMax : ncclSum, c->comm, st));
    return SQGR_OK;
}

int comm_allgather_dev(sqgr_comm* c, const void* dev_send, void* dev_recv, size_t bytes_per_rank, hipStream_t st) {
    if (!c || c->world <= 1) {
        if (dev_send != dev_recv && bytes_per_rank) SQGR_HIP(hipMemcpyAsync(dev_recv, dev_send, bytes_per_rank, hipMemcpyDeviceToDevice, st));
        return SQGR_OK;
    }
    LaunchTimer t(c->ctx, "rccl_allgather", st);
    SQGR_NCCL(g_rccl.AllGather(dev_send, dev_recv, bytes_per_rank, ncclUint8, c->comm, st));
    return SQGR_OK;
}

// Before a data collective every rank tells the others whether its own part succeeded (one all-reduce(max) of a flag): a
// rank that failed — an invalid argument, an allocation — does not enter the data collective and must not leave its peers
// waiting in it.  Returns local_rc if that is an error, SQGR_ERR_HIP if a peer failed, SQGR_OK otherwise.
int comm_agree(sqgr_comm* c, int local_rc, hipStream_t st) {
    if (!c || c->world <= 1) return local_rc;
    int64_t flag = local_rc != SQGR_OK ? 1 : 0;
    const char* keep = nullptr;
    std::string msg;
    if (local_rc != SQGR_OK) msg = sqgr_last_error();  // the agreement below must not overwrite the rank's own message
    int rc = c->stage.ensure(1);
    if (rc == SQGR_OK) {
        hipError_t e = hipMemcpyAsync(c->stage.p, &flag, 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) rc = comm_allreduce_i64_dev(c, c->stage.p, 1, true, st);
        if (e == hipSuccess && rc == SQGR_OK) e = hipMemcpyAsync(&flag, c->stage.p, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && rc == SQGR_OK) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = SQGR_ERR_HIP;
    }
    (void)keep;
    if (local_rc != SQGR_OK) {
        set_error("%s", msg.c_str());
        return local_rc;
    }
    if (rc != SQGR_OK) return rc;
    if (flag) {
        set_error("rank %d: another rank failed before the collective (its own error message says why)", c->rank);
        return SQGR_ERR_HIP;
    }
    return SQGR_OK;
}

int comm_rank(const sqgr_comm* c) { return c ? c->rank : 0; }
int comm_world(const sqgr_comm* c) { return c ? c->world : 1; }

}  // namespace sqgr

extern "C" {

int sqgr_comm_unique_id(uint8_t* out_id) {
    SQGR_REQUIRE(out_id, "out_id is NULL");
    SQGR_TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == SQGR_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    SQGR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return SQGR_OK;
}

int sqgr_comm_create(sqgr_ctx* ctx, const uint8_t* unique_id, int32_t rank, int32_t world, sqgr_comm** out_comm) {
    SQGR_REQUIRE(ctx && unique_id && out_comm, "ctx/unique_id/out_comm is NULL");
    *out_comm = nullptr;
    SQGR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside [0,%d)", rank, world);
    SQGR_TRY(rccl_load());
    SQGR_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    sqgr_comm* c = new sqgr_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return SQGR_ERR_HIP;
    }
    *out_comm = c;
    return SQGR_OK;
}

int sqgr_comm_destroy(sqgr_comm* comm) {
    if (!comm) return SQGR_OK;
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if (comm->comm) (void)g_rccl.CommDestroy(comm->comm);
    delete comm;
    return SQGR_OK;
}

int sqgr_comm_info(const sqgr_comm* comm, int32_t* rank, int32_t* world) {
    SQGR_REQUIRE(comm, "comm is NULL");
    // what RCCL itself says about the communicator, not what the caller passed to sqgr_comm_create (bench.py reports it as `rccl_world`)
    int r = comm->rank, w = comm->world;
    if (comm->comm) {
        SQGR_NCCL(g_rccl.CommUserRank(comm->comm, &r));
        SQGR_NCCL(g_rccl.CommCount(comm->comm, &w));
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return SQGR_OK;
}

int sqgr_comm_allreduce_i64(sqgr_comm* comm, int64_t* buf, int64_t count, int32_t op) {
    SQGR_REQUIRE(comm && (buf || count == 0) && count >= 0, "comm/buf is NULL or count < 0");
    SQGR_REQUIRE(op == SQGR_OP_SUM || op == SQGR_OP_MAX, "op must be SQGR_OP_SUM or SQGR_OP_MAX");
    if (count == 0) return SQGR_OK;
    sqgr_ctx* ctx = comm->ctx;
    SQGR_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SQGR_TRY(comm->stage.ensure((size_t)count));
    SQGR_HIP(hipMemcpyAsync(comm->stage.p, buf, (size_t)count * 8, hipMemcpyHostToDevice, st));
    SQGR_TRY(comm_allreduce_i64_dev(comm, comm->stage.p, (size_t)count, op == SQGR_OP_MAX, st));
    SQGR_HIP(hipMemcpyAsync(buf, comm->stage.p, (size_t)count * 8, hipMemcpyDeviceToHost, st));
    SQGR_HIP(hipStreamSynchronize(st));
    return SQGR_OK;
}

int sqgr_comm_barrier(sqgr_comm* comm) {
    SQGR_REQUIRE(comm, "comm is NULL");
    int64_t one = 1;
    SQGR_TRY(sqgr_comm_allreduce_i64(comm, &one, 1, SQGR_OP_SUM));
    SQGR_REQUIRE(one == comm->world, "barrier all-reduce returned %lld for %d ranks", (long long)one, comm->world);
    return SQGR_OK;
}

}  // extern "C"
