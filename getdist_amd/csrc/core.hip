// Context, device memory, sample upload (SoA transpose), event timers.
#include <stdarg.h>
#include <stdint.h>
#include <ctype.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <string>

#include <map>
#include <mutex>

#include "ctx.hpp"

int gd_fail(gd_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        ctx->err = buf;
        ctx->fetch_pending.clear();  // their destinations may be locals of the entry point that is failing
        ctx->fetch_off = 0;
    }
    return code;
}

__global__ void k_fetch_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}
__global__ void k_fetch_copy4(const unsigned int* __restrict__ src, unsigned int* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}
__global__ void k_fetch_copy1(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}

static int fetch_kernel(gd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return GD_OK;
    const uintptr_t a = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    if ((a & 15) == 0) {
        const size_t n = bytes / 16;
        k_fetch_copy16<<<(unsigned)std::min<size_t>((n + 255) / 256, 1024), 256, 0, ctx->stream>>>((const uint4*)src, (uint4*)dst, n);
    } else if ((a & 3) == 0) {
        const size_t n = bytes / 4;
        k_fetch_copy4<<<(unsigned)std::min<size_t>((n + 255) / 256, 1024), 256, 0, ctx->stream>>>((const unsigned int*)src,
                                                                                             (unsigned int*)dst, n);
    } else {
        k_fetch_copy1<<<(unsigned)std::min<size_t>((bytes + 255) / 256, 1024), 256, 0, ctx->stream>>>((const unsigned char*)src,
                                                                                                (unsigned char*)dst, bytes);
    }
    GD_KERNEL_CHECK();
    return GD_OK;
}

int gd_fetch(gd_ctx* ctx, void* host_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return GD_OK;
    if (!ctx->fetch_block) GD_HIP(hipHostMalloc(&ctx->fetch_block, gd_ctx::kFetchBytes, hipHostMallocDefault));
    const size_t need = (bytes + 15) & ~(size_t)15;
    if (bytes > gd_ctx::kFetchMax || ctx->fetch_off + need > gd_ctx::kFetchBytes) {
        GD_HIP(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, (ctx->stream)));
        return GD_OK;
    }
    const size_t off = ctx->fetch_off;
    ctx->fetch_off += need;
    const int rc = fetch_kernel(ctx, (char*)ctx->fetch_block + off, d_src, bytes);
    if (rc) return rc;
    ctx->fetch_pending.push_back({host_dst, off, bytes});
    return GD_OK;
}

int gd_fetch_pinned(gd_ctx* ctx, void* pinned_dst, const void* d_src, size_t bytes) {
    if (bytes > gd_ctx::kFetchMax) {
        GD_HIP(hipMemcpyAsync(pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, (ctx->stream)));
        return GD_OK;
    }
    return fetch_kernel(ctx, pinned_dst, d_src, bytes);
}

int gd_stream_sync(gd_ctx* ctx) {
    GD_HIP(hipStreamSynchronize((ctx->stream)));
    for (const gd_ctx::Fetch& f : ctx->fetch_pending) memcpy(f.dst, (const char*)ctx->fetch_block + f.off, f.bytes);
    ctx->fetch_pending.clear();
    ctx->fetch_off = 0;
    return GD_OK;
}

int gd_stage_h2d(gd_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    gd_ctx::StageSlot& sl = ctx->stage[ctx->stage_next];
    ctx->stage_next = (ctx->stage_next + 1) % gd_ctx::kStageSlots;
    if (sl.used) GD_HIP(hipEventSynchronize(sl.ev));  // the copy that last read this slot has executed
    if (!ctx->stage_block) {  // one page-locked block serves the usual small tables of all slots (one hipHostMalloc)
        GD_HIP(hipHostMalloc(&ctx->stage_block, (size_t)gd_ctx::kStageSlots * gd_ctx::kStageBytes, hipHostMallocDefault));
        for (int k = 0; k < gd_ctx::kStageSlots; ++k) {
            ctx->stage[k].host = (char*)ctx->stage_block + (size_t)k * gd_ctx::kStageBytes;
            ctx->stage[k].cap = gd_ctx::kStageBytes;
            GD_HIP(hipEventCreateWithFlags(&ctx->stage[k].ev, hipEventDisableTiming));
        }
    }
    if (sl.cap < bytes) {  // an unusually large table: a private block for this slot
        if (sl.own) GD_HIP(hipHostFree(sl.host));
        sl.host = nullptr, sl.own = false;
        GD_HIP(hipHostMalloc(&sl.host, bytes, hipHostMallocDefault));
        sl.cap = bytes, sl.own = true;
    }
    memcpy(sl.host, src, bytes);
    // a copy KERNEL reading the page-locked slot, not a DMA copy: those queue behind the result copies of the previous
    // batched call on the copy engines (measured: a 20-KB table then arrives up to 3.6 ms after it was enqueued, with the
    // stream idle in front of it)
    GD_TRY(fetch_kernel(ctx, d_dst, sl.host, bytes));
    GD_HIP(hipEventRecord(sl.ev, ctx->stream));
    sl.used = true;
    return GD_OK;
}

// Stream-ordered upload of a host table the caller may release on return: small ones through the staging ring above,
// large ones (whole histograms, masks) by DMA.
int gd_h2d(gd_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    if (bytes == 0) return GD_OK;
    if (bytes <= gd_ctx::kStageBytes) return gd_stage_h2d(ctx, d_dst, src, bytes);
    GD_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GD_OK;
}

static int grow(gd_ctx* ctx, void** p, int64_t* have, int64_t bytes) {
    if (bytes <= *have) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    int64_t want = bytes + bytes / 8;
    if (hipMalloc(p, (size_t)want) != hipSuccess) {
        *p = nullptr;
        gd_fail(ctx, GD_ERR_NOMEM, "scratch allocation of %lld bytes failed", (long long)want);
        return -1;
    }
    *have = want;
    return 0;
}
void* gd_scratch(gd_ctx* ctx, int64_t bytes) {
    return grow(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes) ? nullptr : ctx->scratch;
}
void* gd_scratch2(gd_ctx* ctx, int64_t bytes) {
    return grow(ctx, &ctx->scratch2, &ctx->scratch2_bytes, bytes) ? nullptr : ctx->scratch2;
}


// live contexts (a batched call may remember a second context that its owner has destroyed meanwhile)
static std::mutex g_live_mu;
static std::set<gd_ctx*> g_live;
bool gd_ctx_alive(gd_ctx* ctx) {
    std::lock_guard<std::mutex> g(g_live_mu);
    return g_live.count(ctx) != 0;
}

int gd_stream_priority(gd_ctx* ctx, int level) {  // > 0: most urgent, 0: the default, < 0: least urgent
    int least = 0, greatest = 0;
    GD_HIP(hipSetDevice(ctx->device));
    GD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));  // numerically lower = more urgent
    GD_HIP(hipStreamSynchronize(ctx->stream));
    hipStream_t fresh = nullptr;
    const int prio = level > 0 ? greatest : (level < 0 ? least : (least + greatest) / 2);
    GD_HIP(hipStreamCreateWithPriority(&fresh, hipStreamDefault, prio));
    (void)hipStreamDestroy(ctx->stream);
    ctx->stream = fresh;
    return GD_OK;
}

extern "C" {

const char* gd_version(void) { return "gdhip 0.1 (gfx950)"; }

int gd_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int gd_create(int device, gd_ctx** out) {
    if (!out) return GD_ERR_BADARG;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return GD_ERR_NODEVICE;
    if (device < 0 || device >= c) return GD_ERR_BADARG;
    gd_ctx* ctx = new gd_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess ||
        hipStreamCreate(&ctx->copy_stream) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return GD_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    {
        std::lock_guard<std::mutex> g(g_live_mu);
        g_live.insert(ctx);
    }
    *out = ctx;
    return GD_OK;
}

void gd_destroy(gd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->batch_state_release) ctx->batch_state_release(ctx, true);
    gd_comm_release(ctx);
    {
        std::lock_guard<std::mutex> g(g_live_mu);
        g_live.erase(ctx);
    }
    gd_fft_cache_destroy(ctx);
    for (auto& kv : ctx->dctmat) (void)hipFree(kv.second);
    for (auto& kv : ctx->fft_tw) (void)hipFree(kv.second);
    if (ctx->w_sel) ctx->w = ctx->w_main, ctx->w8 = ctx->w8_main;
    if (!ctx->borrowed) {
        if (ctx->cols) (void)hipFree(ctx->cols);
        if (ctx->w) (void)hipFree(ctx->w);
        if (ctx->w8) (void)hipFree(ctx->w8);
    }
    if (ctx->like_w) (void)hipFree(ctx->like_w);
    if (ctx->wcum) (void)hipFree(ctx->wcum);
    ctx->bq.reset();
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch2) (void)hipFree(ctx->scratch2);
    if (ctx->gather_index) (void)hipFree(ctx->gather_index);
    for (auto& ev : ctx->copy_marks)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->kopt_evs)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& sl : ctx->stage) {
        if (sl.own && sl.host) (void)hipHostFree(sl.host);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    if (ctx->stage_block) (void)hipHostFree(ctx->stage_block);
    if (ctx->fetch_block) (void)hipHostFree(ctx->fetch_block);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipEventDestroy(ctx->copy_ev);
    (void)hipStreamDestroy(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* gd_last_error(gd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gd_device_info(gd_ctx* ctx, int64_t* info) {
    GD_REQUIRE(ctx && info, "null argument");
    GD_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    GD_HIP(hipGetDeviceProperties(&prop, ctx->device));
    size_t fr = 0, tot = 0;
    GD_HIP(hipMemGetInfo(&fr, &tot));
    info[0] = prop.multiProcessorCount;
    info[1] = (int64_t)prop.sharedMemPerBlock;
    info[2] = (int64_t)tot;
    info[3] = (int64_t)fr;
    info[4] = prop.clockRate;
    info[5] = prop.warpSize;
    return GD_OK;
}

int gd_sync(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_dev_alloc(gd_ctx* ctx, int64_t bytes, void** d_out) {
    GD_REQUIRE(ctx && d_out && bytes >= 0, "bad argument");
    GD_HIP(hipSetDevice(ctx->device));
    *d_out = nullptr;
    GD_HIP(hipMalloc(d_out, (size_t)(bytes > 0 ? bytes : 8)));
    return GD_OK;
}

int gd_dev_free(gd_ctx* ctx, void* d_ptr) {
    GD_REQUIRE(ctx, "null context");
    if (d_ptr) {
        GD_TRY(gd_stream_sync(ctx));
        GD_HIP(hipFree(d_ptr));
    }
    return GD_OK;
}

int gd_memcpy_h2d(gd_ctx* ctx, void* d_dst, const void* src, int64_t bytes) {
    GD_REQUIRE(ctx && d_dst && src && bytes >= 0, "bad argument");
    GD_HIP(hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_memcpy_d2h(gd_ctx* ctx, void* dst, const void* d_src, int64_t bytes) {
    GD_REQUIRE(ctx && dst && d_src && bytes >= 0, "bad argument");
    GD_TRY(gd_fetch(ctx, dst, d_src, (size_t)bytes));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_memcpy_d2h_async(gd_ctx* ctx, void* dst, const void* d_src, int64_t bytes) {
    GD_REQUIRE(ctx && dst && d_src && bytes >= 0, "bad argument");
    // order after everything queued on the compute stream so far, then copy on the copy stream
    GD_HIP(hipEventRecord(ctx->copy_ev, ctx->stream));
    GD_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_ev, 0));
    GD_HIP(hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    return GD_OK;
}

int gd_copy_sync(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    GD_HIP(hipStreamSynchronize(ctx->copy_stream));
    return GD_OK;
}

int gd_copy_mark(gd_ctx* ctx, int32_t* token_out) {
    GD_REQUIRE(ctx && token_out, "null argument");
    const int slot = ctx->copy_mark_next;
    ctx->copy_mark_next = (slot + 1) % gd_ctx::kCopyMarks;
    if (!ctx->copy_marks[slot]) GD_HIP(hipEventCreateWithFlags(&ctx->copy_marks[slot], hipEventDisableTiming));
    // also order the mark after the compute stream's work so far (status words are copied on that stream)
    GD_HIP(hipEventRecord(ctx->copy_ev, ctx->stream));
    GD_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_ev, 0));
    GD_HIP(hipEventRecord(ctx->copy_marks[slot], ctx->copy_stream));
    *token_out = slot;
    return GD_OK;
}

int gd_copy_wait(gd_ctx* ctx, int32_t token) {
    GD_REQUIRE(ctx && token >= 0 && token < gd_ctx::kCopyMarks && ctx->copy_marks[token], "bad copy mark");
    // a slot re-used by a later mark waits for that later point: the copy stream is FIFO, so that covers this one
    GD_HIP(hipEventSynchronize(ctx->copy_marks[token]));
    return GD_OK;
}

int gd_memcpy_d2d(gd_ctx* ctx, void* d_dst, const void* d_src, int64_t bytes) {
    GD_REQUIRE(ctx && d_dst && d_src && bytes >= 0, "bad argument");
    GD_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return GD_OK;
}

// Large page-locked blocks (the landing blocks of a triangle's grids: 642 MB) on transparent huge pages: anonymous memory
// advised MADV_HUGEPAGE, touched, then registered with the runtime.  hipHostMalloc hands out 4-KB pages; a block allocated late
// in a process comes from wherever the kernel finds them, and its result copies then run a few per cent slower than those
// into a block allocated early (bench.py's delivered triangles alternate between two blocks: 29 / 27 ms in the processes
// where that happened).  GDHIP_HOST_ALLOC_PLAIN=1: hipHostMalloc for every size.
namespace {
struct HugeBlock {
    void* map;
    size_t len;
};
std::mutex g_huge_mu;
std::map<void*, HugeBlock> g_huge;
constexpr size_t kHuge = (size_t)2 << 20;

// the NUMA node the device hangs on (sysfs; -1: unknown / a single node)
int device_numa_node(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}
}  // namespace

int gd_host_alloc(gd_ctx* ctx, int64_t bytes, void** out) {
    GD_REQUIRE(ctx && out && bytes > 0, "bad argument");
    *out = nullptr;
    if ((size_t)bytes >= 16 * kHuge && getenv("GDHIP_HOST_ALLOC_PLAIN") == nullptr) {
        const size_t len = ((size_t)bytes + kHuge - 1) / kHuge * kHuge;
        void* map = mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map != MAP_FAILED) {
            char* p = (char*)(((uintptr_t)map + kHuge - 1) / kHuge * kHuge);
            (void)madvise(p, len, MADV_HUGEPAGE);
            const int node = device_numa_node(ctx->device);
            if (node >= 0 && node < 64) {  // prefer the device's node, as hipHostMalloc does (best effort)
                unsigned long mask = 1UL << node;
                (void)syscall(SYS_mbind, p, len, 1 /* MPOL_PREFERRED */, &mask, sizeof(mask) * 8 + 1, 0);
            }
            for (size_t o = 0; o < len; o += 4096) p[o] = 0;  // first touch: the pages exist before they are pinned
            if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> g(g_huge_mu);
                g_huge[p] = HugeBlock{map, len + kHuge};
                *out = p;
                return GD_OK;
            }
            (void)hipGetLastError();
            munmap(map, len + kHuge);
        }
    }
    GD_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return GD_OK;
}

int gd_host_free(gd_ctx* ctx, void* ptr) {
    GD_REQUIRE(ctx, "null context");
    if (ptr) {
        GD_TRY(gd_stream_sync(ctx));
        HugeBlock hb{nullptr, 0};
        {
            std::lock_guard<std::mutex> g(g_huge_mu);
            auto it = g_huge.find(ptr);
            if (it != g_huge.end()) hb = it->second, g_huge.erase(it);
        }
        if (hb.map) {
            GD_HIP(hipHostUnregister(ptr));
            munmap(hb.map, hb.len);
        } else {
            GD_HIP(hipHostFree(ptr));
        }
    }
    return GD_OK;
}

int gd_memset(gd_ctx* ctx, void* d_dst, int value, int64_t bytes) {
    GD_REQUIRE(ctx && d_dst && bytes >= 0, "bad argument");
    GD_HIP(hipMemsetAsync(d_dst, value, (size_t)bytes, ctx->stream));
    return GD_OK;
}

int gd_timer_start(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    GD_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return GD_OK;
}

int gd_timer_stop_ms(gd_ctx* ctx, double* ms_out) {
    GD_REQUIRE(ctx && ms_out, "null argument");
    GD_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    GD_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    GD_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_out = ms;
    return GD_OK;
}

}  // extern "C"

// ---- upload -----------------------------------------------------------------------------------------
// Row-major (N x n) -> SoA.  32x32 tile transpose through LDS; reads and writes both coalesced.
__global__ void transpose_rows_to_cols(const double* __restrict__ X, int64_t N, int64_t n, int64_t row_stride,
                                       double* __restrict__ cols, int64_t ld) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int64_t r = r0 + k, c = c0 + threadIdx.x;
        if (r < N && c < n) tile[k][threadIdx.x] = X[r * row_stride + c];
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int64_t c = c0 + k, r = r0 + threadIdx.x;
        if (r < N && c < n) cols[c * ld + r] = tile[threadIdx.x][k];
    }
}

__global__ void k_gather_items(double2* __restrict__ dst, const double2* __restrict__ src, const int* __restrict__ index,
                               int64_t item16) {
    const double2* s = src + (int64_t)index[blockIdx.y] * item16;
    double2* d = dst + (int64_t)blockIdx.y * item16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < item16; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

// weights that are all non-negative integers (MCMC multiplicities) allow exact u32 LDS counters in the 2D binning
__global__ void k_weights_integral(const double* __restrict__ w, int64_t N, int* __restrict__ bad, double* __restrict__ sum) {
    double s = 0;
    int b = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = w[i];
        if (!(v >= 0.0) || v != trunc(v) || v > 1048576.0) b |= 1;
        if (v > 255.0) b |= 2;  // too large for the byte copy
        if (!(v >= 0.0)) b |= 4;  // negative or NaN: no fixed-point bucket sums for the quantile select (w_sum stays unknown)
        s += v;
    }
    if (b) atomicOr(bad, b);
    s = wave_sum(s);
    // one slot per wave, added by the host in slot order: the total is the same in every run (the quantile buckets'
    // fixed-point scale is derived from it)
    if ((threadIdx.x & 63) == 0) sum[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
}

__global__ void k_weights_to_u8(const double* __restrict__ w, int64_t N, unsigned char* __restrict__ w8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        w8[i] = (unsigned char)w[i];
}

extern "C" {

int gd_gather_items(gd_ctx* ctx, void* d_dst, const void* d_src, const int32_t* index, int32_t count, int64_t item_bytes) {
    GD_REQUIRE(ctx && d_dst && d_src && index && count > 0 && item_bytes > 0 && item_bytes % 16 == 0, "bad argument");
    // (a block of its own: density2d's periodic route gathers INTO scratch2 -- with the index list at the head of the same
    // block the kernel read its indices from under its own stores, and the second pair of a call came out as noise in ~7 %
    // of the calls)
    if (grow(ctx, &ctx->gather_index, &ctx->gather_index_bytes, (int64_t)count * 4)) return GD_ERR_NOMEM;
    int* d_index = (int*)ctx->gather_index;
    // stream-ordered: returns once enqueued (every consumer of d_dst is an entry point of this context)
    int rc = gd_stage_h2d(ctx, d_index, index, (size_t)count * 4);
    if (rc) return rc;
    int bx = (int)((item_bytes / 16 + 255) / 256);
    if (bx > 64) bx = 64;
    k_gather_items<<<dim3(bx, count), 256, 0, ctx->stream>>>((double2*)d_dst, (const double2*)d_src, d_index, item_bytes / 16);
    GD_KERNEL_CHECK();
    return GD_OK;
}

// Builds the new sample set in locals; the context is only touched once everything has succeeded, so a failed upload
// leaves it EMPTY (cols == nullptr, N == 0), never half-initialised.
// shard_first >= 0: X holds only the columns [shard_first, shard_first + shard_count) of the n (column-major); the others
// are filled in by gd_comm_share_columns
static int upload_build(gd_ctx* ctx, const double* X, int64_t N, int64_t n, int64_t row_stride, int64_t col_stride,
                        const double* weights, int64_t ld, double** cols_out, double** w_out, unsigned char** w8_out,
                        bool* integral_out, double* wsum_out, int64_t shard_first = -1, int64_t shard_count = 0) {
    double*& cols = *cols_out;
    double*& w = *w_out;
    unsigned char*& w8 = *w8_out;
    GD_HIP(hipMalloc((void**)&cols, (size_t)(ld * (n + GD_EXTRA_COLS) * 8)));
    GD_HIP(hipMemsetAsync(cols + ld * n, 0, (size_t)(ld * GD_EXTRA_COLS * 8), ctx->stream));
    if (shard_first >= 0) {
        // (the pad rows of every column are zeroed: the other ranks receive whole ld-row blocks)
        if (ld > N)
            for (int64_t j = 0; j < shard_count; ++j)
                GD_HIP(hipMemsetAsync(cols + (shard_first + j) * ld + N, 0, (size_t)((ld - N) * 8), ctx->stream));
        for (int64_t j = 0; j < shard_count; ++j)
            GD_HIP(hipMemcpyAsync(cols + (shard_first + j) * ld, X + j * col_stride, (size_t)(N * 8), hipMemcpyHostToDevice, ctx->stream));
    } else if (row_stride == 1) {
        // column-major host input: one contiguous copy per column
        for (int64_t j = 0; j < n; ++j)
            GD_HIP(hipMemcpyAsync(cols + j * ld, X + j * col_stride, (size_t)(N * 8), hipMemcpyHostToDevice, ctx->stream));
    } else {
        GD_REQUIRE(col_stride == 1 && row_stride >= n, "samples must be C- or Fortran-contiguous");
        // stage row blocks through scratch and transpose on the device
        const int64_t rows_per = 4 << 20;  // 4M rows x n x 8 B per staged block
        for (int64_t r0 = 0; r0 < N; r0 += rows_per) {
            int64_t nr = (N - r0 < rows_per) ? N - r0 : rows_per;
            double* stage = (double*)gd_scratch(ctx, nr * row_stride * 8);
            if (!stage) return GD_ERR_NOMEM;
            GD_HIP(hipMemcpyAsync(stage, X + r0 * row_stride, (size_t)(nr * row_stride * 8), hipMemcpyHostToDevice,
                                  ctx->stream));
            dim3 grid((unsigned)((nr + 31) / 32), (unsigned)((n + 31) / 32)), block(32, 8);
            transpose_rows_to_cols<<<grid, block, 0, ctx->stream>>>(stage, nr, n, row_stride, cols + r0, ld);
            GD_KERNEL_CHECK();
            GD_TRY(gd_stream_sync(ctx));
        }
    }
    *integral_out = false;
    *wsum_out = 0;
    if (weights) {
        GD_HIP(hipMalloc((void**)&w, (size_t)(ld * 8)));
        GD_HIP(hipMemcpyAsync(w, weights, (size_t)(N * 8), hipMemcpyHostToDevice, ctx->stream));
        constexpr int WSLOTS = 1024 * 4;
        char* chk = (char*)gd_scratch(ctx, 128 + WSLOTS * 8);
        if (!chk) return GD_ERR_NOMEM;
        GD_HIP(hipMemsetAsync(chk, 0, 128, ctx->stream));
        k_weights_integral<<<1024, 256, 0, ctx->stream>>>(w, N, (int*)chk, (double*)(chk + 128));
        GD_KERNEL_CHECK();
        int bad = 1;
        double sum = 0;
        std::vector<double> slots(WSLOTS);
        GD_TRY(gd_fetch(ctx, &bad, chk, 4));
        GD_TRY(gd_fetch(ctx, slots.data(), chk + 128, (size_t)WSLOTS * 8));
        GD_TRY(gd_stream_sync(ctx));
        for (double v : slots) sum += v;
        *wsum_out = (bad & 4) ? 0.0 : sum;
        *integral_out = ((bad & 1) == 0) && sum < 4.0e9;
        if (*integral_out && bad == 0) {  // byte multiplicities for the 16-bit packed 2D binning
            GD_HIP(hipMalloc((void**)&w8, (size_t)ld));
            GD_HIP(hipMemsetAsync(w8, 0, (size_t)ld, ctx->stream));
            k_weights_to_u8<<<1024, 256, 0, ctx->stream>>>(w, N, w8);
            GD_KERNEL_CHECK();
        }
    }
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

static int upload_common(gd_ctx* ctx, const double* X, int64_t N, int64_t n, int64_t row_stride, int64_t col_stride,
                         const double* weights, int64_t shard_first, int64_t shard_count);

int gd_upload(gd_ctx* ctx, const double* X, int64_t N, int64_t n, int64_t row_stride, int64_t col_stride,
              const double* weights) {
    GD_REQUIRE(ctx && X && N > 0 && n > 0, "bad sample array");
    return upload_common(ctx, X, N, n, row_stride, col_stride, weights, -1, 0);
}

int gd_upload_shard(gd_ctx* ctx, const double* X_cols, int64_t N, int64_t n, int64_t col_first, int64_t col_count,
                    int64_t col_stride, const double* weights) {
    GD_REQUIRE(ctx && N > 0 && n > 0 && col_first >= 0 && col_count >= 0 && col_first + col_count <= n, "bad column shard");
    GD_REQUIRE(col_count == 0 || (X_cols && col_stride >= N), "bad column shard");
    return upload_common(ctx, X_cols, N, n, 1, col_stride, weights, col_first, col_count);
}

static int upload_common(gd_ctx* ctx, const double* X, int64_t N, int64_t n, int64_t row_stride, int64_t col_stride,
                         const double* weights, int64_t shard_first, int64_t shard_count) {
    GD_HIP(hipSetDevice(ctx->device));
    GD_TRY(gd_stream_sync(ctx));
    // The stream of a context that OWNS a sample set is urgent: in a batched call it carries the optimiser's stage A, whose
    // small launches (table copies, 128-thread sums) otherwise wait for a wave slot behind the saturating launches of the
    // convolution on the second stream (C3 step 24.2 against 24.8 ms, three alternating runs; the second stream high as
    // well, or low: no difference).  Once, at the first upload -- before any plan or communicator is tied to the stream.
    if (!ctx->main_high && !getenv("GDHIP_MAIN_STREAM_NORMAL")) {
        GD_TRY(gd_stream_priority(ctx, 1));
        ctx->main_high = true;
    }
    if (ctx->batch_state_release) ctx->batch_state_release(ctx, false);  // index columns of the old sample set
    if (ctx->w_sel) ctx->w = ctx->w_main, ctx->w8 = ctx->w8_main;
    if (!ctx->borrowed) {
        if (ctx->cols) (void)hipFree(ctx->cols);
        if (ctx->w) (void)hipFree(ctx->w);
        if (ctx->w8) (void)hipFree(ctx->w8);
    }
    ctx->w8 = ctx->w8_main = nullptr;
    ctx->borrowed = false;
    if (ctx->like_w) (void)hipFree(ctx->like_w);
    if (ctx->wcum) (void)hipFree(ctx->wcum);
    ctx->wcum = nullptr;
    ctx->cols = ctx->w = ctx->like_w = ctx->w_main = nullptr;
    ctx->w_sel = 0;
    ctx->w_integral = false;
    ctx->w_sum = ctx->w_main_sum = 0;
    ctx->N = ctx->n = ctx->ld = 0;
    ctx->bq.reset();
    const int64_t ld = (N + 511) / 512 * 512;
    double *cols = nullptr, *w = nullptr;
    unsigned char* w8 = nullptr;
    bool integral = false;
    double wsum = 0;
    const int rc = upload_build(ctx, X, N, n, row_stride, col_stride, weights, ld, &cols, &w, &w8, &integral, &wsum, shard_first,
                                shard_count);
    if (rc != GD_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        if (cols) (void)hipFree(cols);
        if (w) (void)hipFree(w);
        if (w8) (void)hipFree(w8);
        return rc;
    }
    ctx->cols = cols;
    ctx->w = w;
    ctx->w8 = w8;
    ctx->w_integral = integral;
    ctx->w_sum = wsum;
    ctx->N = N;
    ctx->n = n;
    ctx->ld = ld;
    ctx->bq = std::make_shared<BucketCols>();  // (the old set's bucket columns go with its last user)
    return GD_OK;
}

int gd_num_rows(gd_ctx* ctx, int64_t* N, int64_t* n) {
    GD_REQUIRE(ctx && N && n, "null argument");
    *N = ctx->N;
    *n = ctx->n;
    return GD_OK;
}

int gd_column_ptr(gd_ctx* ctx, int64_t j, void** d_out) {
    GD_REQUIRE(ctx && d_out, "null argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    if (j == -2) {
        *d_out = ctx->w;
        return GD_OK;
    }
    GD_REQUIRE(j >= 0 && j < ctx->n + GD_EXTRA_COLS, "column out of range");
    *d_out = ctx->cols + j * ctx->ld;
    return GD_OK;
}

// like weights = w * exp(mean_loglike - loglikes)  (mode 0; mcsamples.py:1560,1830)  or  w * loglikes  (mode 1; :1558)
__global__ void k_like_weights(const double* __restrict__ w, const double* __restrict__ loglikes, int64_t N, int mode,
                               double mean_loglike, double* __restrict__ out, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double wi = w ? w[i] : 1.0, l = loglikes[i];
        const double v = (mode == 1) ? wi * l : wi * exp(mean_loglike - l);
        out[i] = v;
        s += v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

int gd_like_weights(gd_ctx* ctx, const double* loglikes, int32_t mode, double mean_loglike, double* sum_out) {
    GD_REQUIRE(ctx, "null context");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(ctx->w_sel == 0, "like weights are selected; call gd_select_weights(ctx, 0) first");
    if (!loglikes) {
        if (ctx->like_w) (void)hipFree(ctx->like_w);
        ctx->like_w = nullptr;
        return GD_OK;
    }
    GD_REQUIRE(mode == 0 || mode == 1, "unknown like-weight mode");
    if (!ctx->like_w) {
        GD_HIP(hipMalloc((void**)&ctx->like_w, (size_t)(ctx->ld * 8)));
        GD_HIP(hipMemsetAsync(ctx->like_w, 0, (size_t)(ctx->ld * 8), ctx->stream));
    }
    const int nblk = 2048;
    const int64_t stage_bytes = (ctx->N * 8 + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, stage_bytes + nblk * 8);
    if (!base) return GD_ERR_NOMEM;
    double* stage = (double*)base;
    double* d_part = (double*)(base + stage_bytes);
    GD_HIP(hipMemcpyAsync(stage, loglikes, (size_t)(ctx->N * 8), hipMemcpyHostToDevice, ctx->stream));
    k_like_weights<<<nblk, 256, 0, ctx->stream>>>(ctx->w, stage, ctx->N, mode, mean_loglike, ctx->like_w, d_part);
    GD_KERNEL_CHECK();
    std::vector<double> part((size_t)nblk);
    GD_TRY(gd_fetch(ctx, part.data(), d_part, (size_t)nblk * 8));
    GD_TRY(gd_stream_sync(ctx));
    if (sum_out) {
        double tot = 0;
        for (double v : part) tot += v;
        *sum_out = tot;
    }
    return GD_OK;
}

int gd_select_weights(gd_ctx* ctx, int32_t which) {
    GD_REQUIRE(ctx, "null context");
    GD_REQUIRE(which == 0 || which == 1, "which must be 0 (sample weights) or 1 (like weights)");
    if (which == ctx->w_sel) return GD_OK;
    if (which == 1) {
        GD_REQUIRE(ctx->like_w, "no like weights: call gd_like_weights first");
        ctx->w_main = ctx->w;
        ctx->w_main_integral = ctx->w_integral;
        ctx->w8_main = ctx->w8;
        ctx->w = ctx->like_w;
        ctx->w_integral = false;
        ctx->w_main_sum = ctx->w_sum;
        ctx->w_sum = 0;
        ctx->w8 = nullptr;
    } else {
        ctx->w = ctx->w_main;
        ctx->w_integral = ctx->w_main_integral;
        ctx->w_sum = ctx->w_main_sum;
        ctx->w8 = ctx->w8_main;
        ctx->w_main = nullptr;
        ctx->w8_main = nullptr;
    }
    ctx->w_sel = which;
    return GD_OK;
}

int gd_set_extra_column(gd_ctx* ctx, int32_t slot, const double* x) {
    GD_REQUIRE(ctx && x, "null argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(slot >= 0 && slot < GD_EXTRA_COLS, "extra column slot out of range");
    GD_HIP(hipMemcpyAsync(ctx->cols + (ctx->n + slot) * ctx->ld, x, (size_t)(ctx->N * 8), hipMemcpyHostToDevice,
                          ctx->stream));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_aux_weights(gd_ctx* ctx, const double* w) {
    GD_REQUIRE(ctx && w, "null argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(ctx->w_sel == 0, "auxiliary weights are selected; call gd_select_weights(ctx, 0) first");
    if (!ctx->like_w) {
        GD_HIP(hipMalloc((void**)&ctx->like_w, (size_t)(ctx->ld * 8)));
        GD_HIP(hipMemsetAsync(ctx->like_w, 0, (size_t)(ctx->ld * 8), ctx->stream));
    }
    GD_HIP(hipMemcpyAsync(ctx->like_w, w, (size_t)(ctx->N * 8), hipMemcpyHostToDevice, ctx->stream));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_bind_thread(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    GD_HIP(hipSetDevice(ctx->device));  // the current device is per host thread in HIP
    return GD_OK;
}

int gd_attach_samples(gd_ctx* ctx, gd_ctx* owner) {
    GD_REQUIRE(ctx && owner && ctx != owner, "bad argument");
    GD_REQUIRE(owner->cols && !owner->borrowed, "the owner has no resident sample set of its own");
    GD_REQUIRE(ctx->device == owner->device, "contexts must live on the same device");
    GD_REQUIRE(owner->w_sel == 0, "owner has auxiliary weights selected");
    GD_HIP(hipSetDevice(ctx->device));
    GD_TRY(gd_stream_sync(ctx));
    GD_HIP(hipStreamSynchronize(owner->stream));
    if (ctx->w_sel) ctx->w = ctx->w_main, ctx->w8 = ctx->w8_main, ctx->w_sel = 0;
    if (!ctx->borrowed) {
        if (ctx->cols) (void)hipFree(ctx->cols);
        if (ctx->w) (void)hipFree(ctx->w);
        if (ctx->w8) (void)hipFree(ctx->w8);
    }
    ctx->w8_main = nullptr;
    if (ctx->like_w) (void)hipFree(ctx->like_w);
    if (ctx->wcum) (void)hipFree(ctx->wcum);
    ctx->like_w = ctx->w_main = nullptr;
    ctx->wcum = nullptr;
    ctx->cols = owner->cols;
    ctx->w = owner->w;
    ctx->w8 = owner->w8;
    ctx->w_integral = owner->w_integral;
    ctx->w_sum = owner->w_sum;
    ctx->N = owner->N;
    ctx->n = owner->n;
    ctx->ld = owner->ld;
    ctx->bq = owner->bq;
    ctx->borrowed = true;
    return GD_OK;
}

}  // extern "C"
