// RCCL inside the C ABI: a communicator per context, the two collectives the multi-GPU decomposition of the path needs
// (an all-gather of per-rank vectors -- partial moments, per-parameter state, N_eff -- and a sum
// all-reduce for additive partial tables), issued on the context's stream on device buffers.  librccl.so is resolved
// at the first gd_comm_* call (dlopen), so single-GPU users never load it.
#include <dlfcn.h>

#include "ctx.hpp"

namespace {

// the slice of rccl.h this file uses (RCCL = NCCL's API on ROCm; values from /opt/rocm/include/rccl/rccl.h)
typedef struct {
    char internal[128];
} rcclUniqueId;
typedef void* rcclComm_t;
enum { rcclSuccess = 0, rcclSum = 0, rcclFloat64 = 8 };

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;

const char* load_rccl() {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_rccl.lib) return nullptr;
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return "librccl.so not found";
    Rccl r;
    r.lib = lib;
    *(void**)&r.GetUniqueId = dlsym(lib, "ncclGetUniqueId");
    *(void**)&r.CommInitRank = dlsym(lib, "ncclCommInitRank");
    *(void**)&r.CommDestroy = dlsym(lib, "ncclCommDestroy");
    *(void**)&r.AllGather = dlsym(lib, "ncclAllGather");
    *(void**)&r.AllReduce = dlsym(lib, "ncclAllReduce");
    *(void**)&r.GetErrorString = dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce) return "librccl.so lacks the collective entry points";
    g_rccl = r;
    return nullptr;
}

int rccl_fail(gd_ctx* ctx, const char* what, int rc) {
    return gd_fail(ctx, GD_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}

}  // namespace

void gd_comm_release(gd_ctx* ctx) {  // gd_destroy
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((rcclComm_t)ctx->comm);
    ctx->comm = nullptr;
    if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
    ctx->comm_buf = nullptr;
    ctx->comm_buf_bytes = 0;
}

static int comm_buffer(gd_ctx* ctx, size_t bytes, char** out) {
    if (bytes > ctx->comm_buf_bytes) {
        if (ctx->comm_buf) {
            GD_TRY(gd_stream_sync(ctx));
            (void)hipFree(ctx->comm_buf);
            ctx->comm_buf = nullptr, ctx->comm_buf_bytes = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        GD_HIP(hipMalloc(&ctx->comm_buf, want));
        ctx->comm_buf_bytes = want;
    }
    *out = (char*)ctx->comm_buf;
    return GD_OK;
}

extern "C" {

int gd_comm_unique_id(void* id128_out) {
    if (!id128_out) return GD_ERR_BADARG;
    if (load_rccl()) return GD_ERR_NODEVICE;
    rcclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != rcclSuccess) return GD_ERR_HIP;
    memcpy(id128_out, id.internal, 128);
    return GD_OK;
}

int gd_comm_init(gd_ctx* ctx, int32_t world, int32_t rank, const void* id128) {
    GD_REQUIRE(ctx && id128 && world >= 1 && rank >= 0 && rank < world, "bad argument");
    if (const char* e = load_rccl()) return gd_fail(ctx, GD_ERR_NODEVICE, "%s", e);
    GD_HIP(hipSetDevice(ctx->device));
    if (ctx->comm) gd_comm_release(ctx);
    rcclUniqueId id;
    memcpy(id.internal, id128, 128);
    rcclComm_t comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
    if (rc != rcclSuccess) return rccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->comm_world = world, ctx->comm_rank = rank;
    return GD_OK;
}

int gd_comm_info(gd_ctx* ctx, int32_t* world_out, int32_t* rank_out) {
    GD_REQUIRE(ctx && world_out && rank_out, "null argument");
    *world_out = ctx->comm ? ctx->comm_world : 0;
    *rank_out = ctx->comm ? ctx->comm_rank : 0;
    return GD_OK;
}

int gd_comm_destroy(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    GD_HIP(hipSetDevice(ctx->device));
    GD_TRY(gd_stream_sync(ctx));
    gd_comm_release(ctx);
    return GD_OK;
}

int gd_comm_allgather_dev(gd_ctx* ctx, const void* d_send, int64_t count, void* d_recv) {
    GD_REQUIRE(ctx && ctx->comm && d_send && d_recv && count > 0, "no communicator / bad argument");
    const int rc = g_rccl.AllGather(d_send, d_recv, (size_t)count, rcclFloat64, (rcclComm_t)ctx->comm, ctx->stream);
    if (rc != rcclSuccess) return rccl_fail(ctx, "ncclAllGather", rc);
    return GD_OK;
}

int gd_comm_allreduce_sum_dev(gd_ctx* ctx, const void* d_send, int64_t count, void* d_recv) {
    GD_REQUIRE(ctx && ctx->comm && d_send && d_recv && count > 0, "no communicator / bad argument");
    const int rc = g_rccl.AllReduce(d_send, d_recv, (size_t)count, rcclFloat64, rcclSum, (rcclComm_t)ctx->comm, ctx->stream);
    if (rc != rcclSuccess) return rccl_fail(ctx, "ncclAllReduce", rc);
    return GD_OK;
}

int gd_comm_allgather(gd_ctx* ctx, const double* send, int64_t count, double* recv) {
    GD_REQUIRE(ctx && ctx->comm && send && recv && count > 0, "no communicator / bad argument");
    GD_HIP(hipSetDevice(ctx->device));
    const int W = ctx->comm_world;
    char* buf;
    GD_TRY(comm_buffer(ctx, (size_t)(W + 1) * count * 8, &buf));
    double* d_send = (double*)buf;
    double* d_recv = d_send + count;
    GD_TRY(gd_h2d(ctx, d_send, send, (size_t)count * 8));
    GD_TRY(gd_comm_allgather_dev(ctx, d_send, count, d_recv));
    GD_TRY(gd_fetch(ctx, recv, d_recv, (size_t)W * count * 8));
    return gd_stream_sync(ctx);
}

int gd_comm_allreduce_sum(gd_ctx* ctx, double* inout, int64_t count) {
    GD_REQUIRE(ctx && ctx->comm && inout && count > 0, "no communicator / bad argument");
    GD_HIP(hipSetDevice(ctx->device));
    char* buf;
    GD_TRY(comm_buffer(ctx, (size_t)2 * count * 8, &buf));
    double* d_send = (double*)buf;
    double* d_recv = d_send + count;
    GD_TRY(gd_h2d(ctx, d_send, inout, (size_t)count * 8));
    GD_TRY(gd_comm_allreduce_sum_dev(ctx, d_send, count, d_recv));
    GD_TRY(gd_fetch(ctx, inout, d_recv, (size_t)count * 8));
    return gd_stream_sync(ctx);
}

}  // extern "C"
