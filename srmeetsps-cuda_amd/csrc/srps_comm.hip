// srps_comm.hip -- RCCL behind the C ABI: communicators bound to contexts and the collectives of the
// image-sharded pass, enqueued on the context's stream.  The reference is single-GPU (its only device code is
// cudaSetDevice(Preferences::deviceId), SRPS.cu:88); this is the seam a multi-GPU host belongs to.
//
// librccl is resolved at run time (dlopen), like roctx: libsrps_hip.so loads and runs on one GPU without it, and inside a
// Python process it binds to the copy PyTorch has already loaded (same SONAME) instead of a second one.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include "srps_internal.h"

namespace srps {

namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_resolve() {
    void* h = nullptr;
    const char* env = getenv("SRPS_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { g_rccl.why = "librccl.so not found (set SRPS_RCCL_LIB to its path)"; return; }
    bool all = true;
#define SRPS_SYM(field, name)                                          \
    do {                                                               \
        g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);         \
        if (!g_rccl.field) { all = false; g_rccl.why = std::string("librccl does not export ") + name; } \
    } while (0)
    SRPS_SYM(GetUniqueId, "ncclGetUniqueId"); SRPS_SYM(CommInitRank, "ncclCommInitRank"); SRPS_SYM(CommInitAll, "ncclCommInitAll");
    SRPS_SYM(CommDestroy, "ncclCommDestroy"); SRPS_SYM(CommCount, "ncclCommCount"); SRPS_SYM(CommUserRank, "ncclCommUserRank");
    SRPS_SYM(AllReduce, "ncclAllReduce"); SRPS_SYM(Broadcast, "ncclBroadcast"); SRPS_SYM(Send, "ncclSend"); SRPS_SYM(Recv, "ncclRecv");
    SRPS_SYM(GroupStart, "ncclGroupStart"); SRPS_SYM(GroupEnd, "ncclGroupEnd"); SRPS_SYM(GetErrorString, "ncclGetErrorString");
#undef SRPS_SYM
    g_rccl.ok = all;
}

int rccl_need() {
    std::call_once(g_rccl_once, rccl_resolve);
    SRPS_REQUIRE(g_rccl.ok, SRPS_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    return SRPS_OK;
}

int rccl_fail(ncclResult_t r, const char* what) {
    set_error("RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
    return SRPS_ERR_HIP;
}
#define SRPS_RCCL(expr)                                        \
    do {                                                       \
        ncclResult_t r_ = (expr);                              \
        if (r_ != ncclSuccess) return rccl_fail(r_, #expr);    \
    } while (0)
}  // namespace

bool comm_bound(const srps_ctx* ctx) { return ctx->comm != nullptr || ctx->host_allreduce != nullptr; }

bool forced_failure(const char* stage) {
    const char* e = getenv("SRPS_FORCE_FAIL");
    if (!e || !*e) return false;
    const size_t n = strlen(stage);
    for (const char* p = e; *p;) {                           // comma-separated, whole words
        const char* q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len == n && !strncmp(p, stage, n)) return true;
        if (!q) break;
        p = q + 1;
    }
    return false;
}

// the caller's collectives (srps_set_host_collectives) instead of RCCL: host functions on device pointers, called with the
// stream(s) that produced the data drained
static int hosted_all_reduce(srps_ctx* ctx, hipStream_t st, void* d_buf, size_t n, int f64) {
    SRPS_HIP(hipStreamSynchronize(st));
    const int rc = ctx->host_allreduce(ctx->host_user, d_buf, n, f64);
    SRPS_REQUIRE(rc == 0, SRPS_ERR_HIP, "host collectives: the caller's all-reduce returned %d", rc);
    return SRPS_OK;
}

int comm_all_reduce_sum(srps_ctx* ctx, float* d_buf, size_t n) {
    if (n == 0) return SRPS_OK;
    if (ctx->host_allreduce) return hosted_all_reduce(ctx, ctx->stream, d_buf, n, 0);
    SRPS_REQUIRE(ctx->comm != nullptr, SRPS_ERR_STATE, "all_reduce: no communicator bound to the context (srps_comm_init_rank / srps_comm_init_all / srps_set_comm)");
    SRPS_RCCL(g_rccl.AllReduce(d_buf, d_buf, n, ncclFloat32, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return SRPS_OK;
}
// several pieces of one buffer, as one group, on a stream of the caller's choice (the overlapped exchange of srps_execute_sharded)
int comm_all_reduce_pieces_on(srps_ctx* ctx, hipStream_t st, float* const* d_piece, const size_t* n, int pieces) {
    if (ctx->host_allreduce) {
        for (int k = 0; k < pieces; ++k)
            if (n[k]) SRPS_TRY(hosted_all_reduce(ctx, st, d_piece[k], n[k], 0));
        return SRPS_OK;
    }
    SRPS_REQUIRE(ctx->comm != nullptr, SRPS_ERR_STATE, "all_reduce: no communicator bound to the context");
    SRPS_RCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int k = 0; k < pieces && r == ncclSuccess; ++k)
        if (n[k]) r = g_rccl.AllReduce(d_piece[k], d_piece[k], n[k], ncclFloat32, ncclSum, (ncclComm_t)ctx->comm, st);
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
    SRPS_RCCL(e);
    return SRPS_OK;
}
int comm_broadcast(srps_ctx* ctx, float* d_buf, size_t n, int root) {
    if (n == 0) return SRPS_OK;
    if (ctx->host_broadcast) {
        SRPS_HIP(hipStreamSynchronize(ctx->stream));
        const int rc = ctx->host_broadcast(ctx->host_user, d_buf, n, root);
        SRPS_REQUIRE(rc == 0, SRPS_ERR_HIP, "host collectives: the caller's broadcast returned %d", rc);
        return SRPS_OK;
    }
    SRPS_REQUIRE(ctx->comm != nullptr, SRPS_ERR_STATE, "broadcast: no communicator bound to the context");
    SRPS_RCCL(g_rccl.Broadcast(d_buf, d_buf, n, ncclFloat32, root, (ncclComm_t)ctx->comm, ctx->stream));
    return SRPS_OK;
}
int comm_all_reduce_sum_f64(srps_ctx* ctx, const double* d_in, double* d_out, size_t n) {
    SRPS_REQUIRE(ctx->comm != nullptr, SRPS_ERR_STATE, "all_reduce: no communicator bound to the context");
    if (n == 0) return SRPS_OK;
    SRPS_RCCL(g_rccl.AllReduce(d_in, d_out, n, ncclFloat64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return SRPS_OK;
}
// The halo exchange of the strip-partitioned CG: for each of `nbuf` buffers, n floats to and from the left neighbour (rank
// `left`, < 0: none) and the right neighbour, all as ONE group (sends and receives that wait for each other must be in flight
// together).
int comm_exchange(srps_ctx* ctx, int nbuf, const float* const* send_left, float* const* recv_left, int left,
                  const float* const* send_right, float* const* recv_right, int right, size_t n) {
    SRPS_REQUIRE(ctx->comm != nullptr, SRPS_ERR_STATE, "exchange: no communicator bound to the context");
    if (n == 0 || nbuf == 0 || (left < 0 && right < 0)) return SRPS_OK;
    SRPS_RCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int b = 0; b < nbuf; ++b) {
        if (left >= 0 && r == ncclSuccess) r = g_rccl.Send(send_left[b], n, ncclFloat32, left, (ncclComm_t)ctx->comm, ctx->stream);
        if (left >= 0 && r == ncclSuccess) r = g_rccl.Recv(recv_left[b], n, ncclFloat32, left, (ncclComm_t)ctx->comm, ctx->stream);
        if (right >= 0 && r == ncclSuccess) r = g_rccl.Send(send_right[b], n, ncclFloat32, right, (ncclComm_t)ctx->comm, ctx->stream);
        if (right >= 0 && r == ncclSuccess) r = g_rccl.Recv(recv_right[b], n, ncclFloat32, right, (ncclComm_t)ctx->comm, ctx->stream);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "ncclSend / ncclRecv");
    SRPS_RCCL(e);
    return SRPS_OK;
}
// every rank's piece of a buffer to every rank: piece q = [offset[q], offset[q] + count[q]) floats of d_buf, broadcast from rank q
// (the pieces differ in size by up to one block column, which ncclAllGather does not take)
int comm_all_gather_pieces(srps_ctx* ctx, float* d_buf, const size_t* offset, const size_t* count) {
    SRPS_REQUIRE(ctx->comm != nullptr, SRPS_ERR_STATE, "all_gather: no communicator bound to the context");
    SRPS_RCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int q = 0; q < ctx->comm_world && r == ncclSuccess; ++q)
        if (count[q]) r = g_rccl.Broadcast(d_buf + offset[q], d_buf + offset[q], count[q], ncclFloat32, q, (ncclComm_t)ctx->comm, ctx->stream);
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "ncclBroadcast");
    SRPS_RCCL(e);
    return SRPS_OK;
}

void comm_release(srps_ctx* ctx) {
    if (ctx->comm && ctx->comm_owned && g_rccl.ok) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_owned = false; ctx->comm_rank = 0; ctx->comm_world = 1;
    ctx->host_allreduce = nullptr; ctx->host_broadcast = nullptr; ctx->host_user = nullptr;      // one way of reaching the other ranks at a time
}

}  // namespace srps

using namespace srps;

extern "C" {

int srps_comm_unique_id(void* id) {
    SRPS_REQUIRE(id != nullptr, SRPS_ERR_INVALID, "comm_unique_id: id is NULL");
    static_assert(SRPS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "srps.h: SRPS_COMM_ID_BYTES must be ncclUniqueId's size");
    SRPS_TRY(rccl_need());
    ncclUniqueId u;
    SRPS_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SRPS_OK;
}

int srps_comm_init_rank(srps_ctx* ctx, const void* id, int rank, int world) {
    SRPS_REQUIRE(ctx != nullptr && id != nullptr, SRPS_ERR_INVALID, "comm_init_rank: null argument");
    SRPS_REQUIRE(world >= 1 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "comm_init_rank: rank %d of %d", rank, world);
    SRPS_REQUIRE(!forced_failure("comm"), SRPS_ERR_UNSUPPORTED, "comm_init_rank: refused on purpose (SRPS_FORCE_FAIL=comm)");
    SRPS_TRY(rccl_need());
    SRPS_HIP(hipSetDevice(ctx->device));
    comm_release(ctx);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    SRPS_RCCL(g_rccl.CommInitRank(&c, world, u, rank));
    ctx->comm = c; ctx->comm_owned = true; ctx->comm_rank = rank; ctx->comm_world = world;
    return SRPS_OK;
}

int srps_comm_init_all(srps_ctx* const* ctxs, int n) {
    SRPS_REQUIRE(ctxs != nullptr && n >= 1 && n <= 64, SRPS_ERR_INVALID, "comm_init_all: bad arguments");
    std::vector<int> dev(n);
    for (int i = 0; i < n; ++i) {
        SRPS_REQUIRE(ctxs[i] != nullptr, SRPS_ERR_INVALID, "comm_init_all: context %d is NULL", i);
        dev[i] = ctxs[i]->device;
        for (int k = 0; k < i; ++k)
            SRPS_REQUIRE(dev[k] != dev[i], SRPS_ERR_INVALID, "comm_init_all: contexts %d and %d are both on device %d (RCCL takes one rank per device)", k, i, dev[i]);
    }
    SRPS_REQUIRE(!forced_failure("comm"), SRPS_ERR_UNSUPPORTED, "comm_init_all: refused on purpose (SRPS_FORCE_FAIL=comm)");
    SRPS_TRY(rccl_need());
    std::vector<ncclComm_t> comms(n, nullptr);
    SRPS_RCCL(g_rccl.CommInitAll(comms.data(), n, dev.data()));
    for (int i = 0; i < n; ++i) {
        comm_release(ctxs[i]);
        ctxs[i]->comm = comms[i]; ctxs[i]->comm_owned = true; ctxs[i]->comm_rank = i; ctxs[i]->comm_world = n;
    }
    return SRPS_OK;
}

int srps_set_comm(srps_ctx* ctx, void* rccl_comm, int rank, int world) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "set_comm: null context");
    if (rccl_comm == nullptr) { comm_release(ctx); return SRPS_OK; }
    SRPS_REQUIRE(world >= 1 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "set_comm: rank %d of %d", rank, world);
    SRPS_TRY(rccl_need());
    // the communicator must be one of the RCCL copy this library resolved, and say the same about itself
    int cw = 0, cr = -1;
    SRPS_RCCL(g_rccl.CommCount((ncclComm_t)rccl_comm, &cw));
    SRPS_RCCL(g_rccl.CommUserRank((ncclComm_t)rccl_comm, &cr));
    SRPS_REQUIRE(cw == world && cr == rank, SRPS_ERR_INVALID, "set_comm: the communicator is rank %d of %d, the caller says %d of %d", cr, cw, rank, world);
    comm_release(ctx);
    ctx->comm = rccl_comm; ctx->comm_owned = false; ctx->comm_rank = rank; ctx->comm_world = world;
    return SRPS_OK;
}

int srps_comm_release(srps_ctx* ctx) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "comm_release: null context");
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    return SRPS_OK;
}

int srps_comm_info(srps_ctx* ctx, int* rank, int* world) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "comm_info: null context");
    const bool any = ctx->comm != nullptr || ctx->host_allreduce != nullptr;
    if (ctx->comm != nullptr) {
        // what the communicator says about itself (ncclCommCount / ncclCommUserRank), not what the context was told
        int cw = 0, cr = -1;
        SRPS_RCCL(g_rccl.CommCount((ncclComm_t)ctx->comm, &cw));
        SRPS_RCCL(g_rccl.CommUserRank((ncclComm_t)ctx->comm, &cr));
        SRPS_REQUIRE(cw == ctx->comm_world && cr == ctx->comm_rank, SRPS_ERR_STATE,
                     "comm_info: the communicator is rank %d of %d, the context holds %d of %d", cr, cw, ctx->comm_rank, ctx->comm_world);
    }
    if (rank) *rank = any ? ctx->comm_rank : 0;
    if (world) *world = any ? ctx->comm_world : 0;      // 0: no communicator bound
    return SRPS_OK;
}

// Collectives of the caller's instead of RCCL for the image-sharded pass (MPI, gloo, ...): host functions on DEVICE pointers,
// called with the stream drained; each returns 0 once its reads and writes are complete.
int srps_set_host_collectives(srps_ctx* ctx, int rank, int world, srps_host_allreduce_fn allreduce, srps_host_broadcast_fn broadcast, void* user) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "set_host_collectives: null context");
    if (!allreduce && !broadcast) {
        ctx->host_allreduce = nullptr; ctx->host_broadcast = nullptr; ctx->host_user = nullptr;
        if (!ctx->comm) { ctx->comm_rank = 0; ctx->comm_world = 1; }
        return SRPS_OK;
    }
    SRPS_REQUIRE(allreduce && broadcast, SRPS_ERR_INVALID, "set_host_collectives: both functions, or none");
    SRPS_REQUIRE(world >= 1 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "set_host_collectives: rank %d of %d", rank, world);
    SRPS_REQUIRE(ctx->comm == nullptr, SRPS_ERR_STATE, "set_host_collectives: the context has an RCCL communicator (srps_comm_release first)");
    (void)hipSetDevice(ctx->device);
    ctx->host_allreduce = allreduce; ctx->host_broadcast = broadcast; ctx->host_user = user;
    ctx->comm_rank = rank; ctx->comm_world = world;
    return SRPS_OK;
}

int srps_all_reduce(srps_ctx* ctx, const char* which) {
    SRPS_REQUIRE(ctx != nullptr && which != nullptr, SRPS_ERR_INVALID, "all_reduce: null argument");
    SRPS_HIP(hipSetDevice(ctx->device));
    void* p = nullptr; size_t n = 0;
    SRPS_TRY(srps_exchange(ctx, which, &p, &n));
    return comm_all_reduce_sum(ctx, (float*)p, n);
}

}  // extern "C"
