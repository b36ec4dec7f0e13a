// RCCL inside the C ABI: a communicator per context and the collectives the multi-GPU decomposition of the path needs
// (an all-gather of per-rank vectors -- partial moments, per-parameter state, chain moments --, a sum
// all-reduce for additive partial tables and the N_eff values, and the distribution of column shards of the sample set over
// xGMI), issued on the context's stream on device buffers.
//
// Robustness (round 5; none of this had run with more than one rank on hardware):
//  * ONE RCCL per process: the library is looked up among the objects the process has already mapped first
//    (dlopen(RTLD_NOLOAD): PyTorch's bundled librccl.so when the host application is torch.distributed), and only
//    loaded from the ROCm installation when nobody has; gd_comm_rccl_path reports which file serves.
//  * ncclCommInitRank runs on a helper thread under a watchdog (GDHIP_COMM_TIMEOUT_S, default 120 s): a rank that never
//    arrives makes gd_comm_init return an error instead of hanging the job, and the caller's all-or-nothing agreement
//    (parallel.init_library_comm) falls back to the host application's collectives.  A late-returning helper aborts the
//    communicator it obtained.
//  * the host-vector collectives do not block in hipStreamSynchronize: they poll the stream together with
//    ncclCommGetAsyncError and give up after the same timeout (ncclCommAbort, communicator dropped), so a dead peer is an
//    error code and not a hang inside the library.
//  * Round 6: a wait that gives up drains the stream (bounded) and forgets the pending result deliveries before the error is
//    returned -- nothing is written to the caller's vectors afterwards; steady-state collectives have their own, much larger
//    limit (GDHIP_COMM_STEADY_TIMEOUT_S) and timeouts their own status code (GD_ERR_TIMEOUT); gd_comm_abandon lets a caller
//    whose own watchdog gave up on gd_comm_init make sure a late success installs nothing.
//  * GDHIP_COMM_INJECT_WAIT_TIMEOUT (test hook): the next host-vector collective's wait gives up at once.
//  * GDHIP_COMM_INJECT_HANG_MS (test hook): the helper thread sleeps that long before it joins -- the watchdog path can be
//    exercised with a single rank on a single GPU (tests/test_gpu_rccl_smoke.py).
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

#include "ctx.hpp"

namespace {

// the slice of rccl.h this file uses (RCCL = NCCL's API on ROCm; values from /opt/rocm/include/rccl/rccl.h)
typedef struct {
    char internal[128];
} rcclUniqueId;
typedef void* rcclComm_t;
enum { rcclSuccess = 0, rcclInProgress = 7, rcclSum = 0, rcclFloat64 = 8 };

struct Rccl {
    void* lib = nullptr;
    bool preloaded = false;  // found among the objects already mapped (the host application's RCCL)
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*CommAbort)(rcclComm_t) = nullptr;                 // optional
    int (*CommGetAsyncError)(rcclComm_t, int*) = nullptr;   // optional
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;  // optional (column shards)
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;

const char* load_rccl() {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_rccl.lib) return nullptr;
    static const char* names[] = {"librccl.so", "librccl.so.1"};
    void* lib = nullptr;
    bool preloaded = false;
    for (const char* nm : names)  // the RCCL this process already holds, if any: never a second copy beside it
        if (!lib && (lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) preloaded = true;
    for (const char* nm : names)
        if (!lib) lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return "librccl.so not found";
    Rccl r;
    r.lib = lib;
    r.preloaded = preloaded;
    *(void**)&r.GetUniqueId = dlsym(lib, "ncclGetUniqueId");
    *(void**)&r.CommInitRank = dlsym(lib, "ncclCommInitRank");
    *(void**)&r.CommDestroy = dlsym(lib, "ncclCommDestroy");
    *(void**)&r.CommAbort = dlsym(lib, "ncclCommAbort");
    *(void**)&r.CommGetAsyncError = dlsym(lib, "ncclCommGetAsyncError");
    *(void**)&r.AllGather = dlsym(lib, "ncclAllGather");
    *(void**)&r.AllReduce = dlsym(lib, "ncclAllReduce");
    *(void**)&r.Broadcast = dlsym(lib, "ncclBroadcast");
    *(void**)&r.GroupStart = dlsym(lib, "ncclGroupStart");
    *(void**)&r.GroupEnd = dlsym(lib, "ncclGroupEnd");
    *(void**)&r.GetErrorString = dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce) {
        dlclose(lib);  // (drops the reference this call took; an object the process had mapped stays mapped)
        return "librccl.so lacks the collective entry points";
    }
    g_rccl = r;
    return nullptr;
}

int rccl_fail(gd_ctx* ctx, const char* what, int rc) {
    return gd_fail(ctx, GD_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}

double comm_timeout_s() {
    const char* e = getenv("GDHIP_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0 ? v : 120.0;
}

// The set-up limit above is for "does the peer exist at all".  A collective of a RUNNING job (the N_eff all-reduce inside
// gd_density2d_batch, the chain moments) may legitimately wait long for a rank that is late -- a long N_eff route, chain
// loading, load imbalance -- and aborting the communicator there leaves the other ranks without a way to renegotiate: the
// steady-state limit is separate and much larger (GDHIP_COMM_STEADY_TIMEOUT_S, default 3600 s; 0 or negative = the set-up limit).
double comm_steady_timeout_s() {
    const char* e = getenv("GDHIP_COMM_STEADY_TIMEOUT_S");
    if (!e) return 3600.0;
    const double v = atof(e);
    return v > 0 ? v : comm_timeout_s();
}

// After a communicator was aborted the stream still holds what was queued behind the dead collective (the copy kernels of
// gd_fetch, a DMA copy into the caller's vector for a large result).  Give it a bounded time to run dry so that nothing is
// written to a host buffer after this entry point has returned its error; then the pending deliveries are forgotten.
void drain_after_abort(gd_ctx* ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(ctx->stream) == hipErrorNotReady &&
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0)
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    ctx->fetch_pending.clear();  // (their destinations belong to the caller that is about to see the error)
    ctx->fetch_off = 0;
}

// a communicator that can no longer be trusted (timeout, asynchronous error): abort it -- ncclCommDestroy would wait for
// the peers -- and drop it from the context; the device staging block stays (a later gd_comm_init reuses it)
void comm_drop(gd_ctx* ctx) {
    if (ctx->comm) {
        if (g_rccl.CommAbort)
            g_rccl.CommAbort((rcclComm_t)ctx->comm);
        else if (g_rccl.CommDestroy)
            g_rccl.CommDestroy((rcclComm_t)ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_world = ctx->comm_rank = 0;
    ctx->comm_completed = 0;
    drain_after_abort(ctx);
}

// Wait for everything enqueued on the context's stream WITHOUT blocking in the driver: poll the stream and the
// communicator's asynchronous error state; a dead peer or a rank that never arrives ends in an error after the timeout.
int comm_wait(gd_ctx* ctx, const char* what, bool setup = false) {
    const auto t0 = std::chrono::steady_clock::now();
    // the first host-vector collective after gd_comm_init is the set-up's test exchange; the column broadcasts are set-up too
    const bool steady = !setup && ctx->comm_completed > 0;
    double limit = steady ? comm_steady_timeout_s() : comm_timeout_s();
    if (const char* inj = getenv("GDHIP_COMM_INJECT_WAIT_TIMEOUT"))  // test hook: this wait gives up at once
        if (atoi(inj) > 0) limit = -1.0;
    for (int spin = 0;; ++spin) {
        const hipError_t q = limit < 0 ? hipErrorNotReady : hipStreamQuery(ctx->stream);
        if (q == hipSuccess) {
            ctx->comm_completed++;
            return gd_stream_sync(ctx);  // (drained: returns at once; delivers the staged result vectors)
        }
        if (q != hipErrorNotReady) {
            comm_drop(ctx);
            return gd_fail(ctx, GD_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        }
        if (g_rccl.CommGetAsyncError && ctx->comm && (spin & 63) == 63) {
            int async = rcclSuccess;
            const int rc = g_rccl.CommGetAsyncError((rcclComm_t)ctx->comm, &async);
            if (rc != rcclSuccess || (async != rcclSuccess && async != rcclInProgress)) {
                const int bad = rc != rcclSuccess ? rc : async;
                comm_drop(ctx);
                return rccl_fail(ctx, what, bad);
            }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            comm_drop(ctx);
            return gd_fail(ctx, GD_ERR_TIMEOUT, "%s: no completion after %.0f s (%s): a peer rank is missing, dead or very late; the communicator was aborted",
                           what, limit < 0 ? 0.0 : limit, steady ? "GDHIP_COMM_STEADY_TIMEOUT_S" : "GDHIP_COMM_TIMEOUT_S");
        }
        if (spin < 2000)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// ncclCommInitRank under a watchdog
struct InitJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    int rc = 0;
    rcclComm_t comm = nullptr;
};

}  // namespace

void gd_comm_release(gd_ctx* ctx) {  // gd_destroy
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((rcclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_completed = 0;
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

int gd_comm_rccl_path(char* buf, int32_t len, int32_t* preloaded_out) {
    if (!buf || len <= 0) return GD_ERR_BADARG;
    buf[0] = 0;
    if (load_rccl()) return GD_ERR_NODEVICE;
    Dl_info info;
    if (dladdr((void*)g_rccl.AllReduce, &info) && info.dli_fname) snprintf(buf, (size_t)len, "%s", info.dli_fname);
    if (preloaded_out) *preloaded_out = g_rccl.preloaded ? 1 : 0;
    return GD_OK;
}

int gd_comm_init(gd_ctx* ctx, int32_t world, int32_t rank, const void* id128) {
    GD_REQUIRE(ctx && id128 && world >= 1 && rank >= 0 && rank < world, "bad argument");
    if (const char* e = load_rccl()) return gd_fail(ctx, GD_ERR_NODEVICE, "%s", e);
    GD_HIP(hipSetDevice(ctx->device));
    if (ctx->comm) {  // a communicator is replaced only once its work has drained
        GD_TRY(gd_stream_sync(ctx));
        if (g_rccl.CommDestroy) g_rccl.CommDestroy((rcclComm_t)ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_world = ctx->comm_rank = 0;
    }
    const int abandon_gen = ctx->comm_abandon_gen.load();
    ctx->comm_completed = 0;
    auto job = std::make_shared<InitJob>();
    rcclUniqueId id;
    memcpy(id.internal, id128, 128);
    const int device = ctx->device;
    const char* inject = getenv("GDHIP_COMM_INJECT_HANG_MS");
    const int hang_ms = inject ? atoi(inject) : 0;
    std::thread helper([job, id, world, rank, device, hang_ms] {
        (void)hipSetDevice(device);
        if (hang_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(hang_ms));
        rcclComm_t comm = nullptr;
        const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
        std::unique_lock<std::mutex> lk(job->mu);
        if (job->abandoned) {  // the caller gave up: nobody will ever use this communicator
            lk.unlock();
            if (rc == rcclSuccess && comm) {
                if (g_rccl.CommAbort)
                    g_rccl.CommAbort(comm);
                else
                    g_rccl.CommDestroy(comm);
            }
            return;
        }
        job->rc = rc, job->comm = comm, job->done = true;
        job->cv.notify_all();
    });
    const double limit = comm_timeout_s();
    {
        std::unique_lock<std::mutex> lk(job->mu);
        if (!job->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->done; })) {
            job->abandoned = true;
            lk.unlock();
            helper.detach();
            return gd_fail(ctx, GD_ERR_TIMEOUT, "ncclCommInitRank: rank %d of %d did not join within %.0f s (GDHIP_COMM_TIMEOUT_S)", (int)rank,
                           (int)world, limit);
        }
    }
    helper.join();
    if (job->rc != rcclSuccess) return rccl_fail(ctx, "ncclCommInitRank", job->rc);
    if (ctx->comm_abandon_gen.load() != abandon_gen) {  // the caller's own watchdog gave up on this call (gd_comm_abandon)
        if (g_rccl.CommAbort)
            g_rccl.CommAbort(job->comm);
        else
            g_rccl.CommDestroy(job->comm);
        return gd_fail(ctx, GD_ERR_TIMEOUT, "gd_comm_init: abandoned by the caller while it ran; no communicator installed");
    }
    ctx->comm = job->comm;
    ctx->comm_world = world, ctx->comm_rank = rank;
    return GD_OK;
}

int gd_comm_abandon(gd_ctx* ctx) {
    if (!ctx) return GD_ERR_BADARG;
    ctx->comm_abandon_gen.fetch_add(1);  // nothing else of the context is touched: this may run beside a stuck gd_comm_* call
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
    return comm_wait(ctx, "ncclAllGather");
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
    return comm_wait(ctx, "ncclAllReduce");
}

// Sample distribution over xGMI (SURVEY.md 8e "broadcast once"): after gd_upload_shard every rank holds the columns
// [first[r], first[r + 1]) of ITS rank r; each rank broadcasts its block to the others, all W broadcasts in one RCCL
// group on the context's stream (7 links x ~153 GB/s per GPU against ~50 GB/s of host link: C5's 80 GB per GPU arrive in
// about a tenth of the time W full uploads take, and a host process need only hold its own columns).
int gd_comm_share_columns(gd_ctx* ctx, const int64_t* first_by_rank) {
    GD_REQUIRE(ctx && ctx->comm && first_by_rank, "no communicator / bad argument");
    GD_REQUIRE(ctx->cols && !ctx->borrowed, "no sample set of this context's own");
    GD_REQUIRE(g_rccl.Broadcast && g_rccl.GroupStart && g_rccl.GroupEnd, "librccl.so lacks ncclBroadcast / ncclGroupStart");
    GD_HIP(hipSetDevice(ctx->device));
    const int W = ctx->comm_world;
    GD_REQUIRE(first_by_rank[0] == 0 && first_by_rank[W] == ctx->n, "column blocks must cover [0, n)");
    for (int r = 0; r < W; ++r) GD_REQUIRE(first_by_rank[r] <= first_by_rank[r + 1], "column blocks must be ascending");
    int rc = g_rccl.GroupStart();
    if (rc != rcclSuccess) return rccl_fail(ctx, "ncclGroupStart", rc);
    for (int r = 0; r < W && rc == rcclSuccess; ++r) {
        const int64_t cnt = (first_by_rank[r + 1] - first_by_rank[r]) * ctx->ld;
        if (cnt == 0) continue;
        double* blk = ctx->cols + first_by_rank[r] * ctx->ld;
        rc = g_rccl.Broadcast(blk, blk, (size_t)cnt, rcclFloat64, r, (rcclComm_t)ctx->comm, ctx->stream);
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc != rcclSuccess) return rccl_fail(ctx, "ncclBroadcast", rc);
    if (rc_end != rcclSuccess) return rccl_fail(ctx, "ncclGroupEnd", rc_end);
    return comm_wait(ctx, "ncclBroadcast (column shards)", true);
}

}  // extern "C"
