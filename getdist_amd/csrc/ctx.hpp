// Internal header of libgdhip.so (gfx950 only).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/gdhip.h"

struct FftPlanCache;  // density2d.hip

// Bucket columns (stats.hip, round 6): the 15-bit (weighted: 14-bit) linear bucket index of every sample of a column, written
// by the counting pass of the quantile select while it has the fp64 value in a register.  Every later pass whose result is
// a function of "which interval is x in" reads these 2 bytes instead of the 8-byte sample: the collect pass of the select
// (live buckets), the pre-binning of the 2D / 1D index columns (a bucket that lies inside one bin maps through a table;
// the ~1 % of buckets that straddle a bin edge re-read the sample and take the exact fp64 route, so the indices stay
// bit-exact).  One object per resident sample set, shared by the contexts that borrow it (gd_attach_samples).
struct BucketCols {
    std::mutex mu;
    unsigned short* buf = nullptr;  // [n][ld]
    int64_t n = 0, ld = 0;
    bool tried = false;             // allocation attempted (a failure is not retried for this sample set)
    std::vector<double> mn, scale;  // per column: bucket = clamp((int)((x - mn) * scale), 0, nb - 1)
    std::vector<int> nb;            // 0 = the column's buckets are not valid
    ~BucketCols() {
        if (buf) (void)hipFree(buf);
    }
};

#define GD_EXTRA_COLS 4  // spare columns behind the sample columns (gd_set_extra_column): loglikes, derived vectors

struct gd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // D2H of finished grids overlaps the next batch's kernels
    hipEvent_t copy_ev = nullptr;
    bool main_high = false;  // the compute stream was re-created with high priority at the first upload (core.hip)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cu_count = 256;
    std::string err;
    // sample set: SoA, column j at cols + j*ld, ld = N rounded up to 512 elements
    double* cols = nullptr;
    double* w = nullptr;  // nullptr => unit weights
    bool w_integral = false;  // all weights are non-negative integers with sum < 2^32 (MCMC multiplicities)
    unsigned char* w8 = nullptr;       // the same multiplicities as bytes when none exceeds 255 (k_hist2d_u16)
    unsigned char* w8_main = nullptr;  // parked while auxiliary weights are selected
    // mean-likelihood weights (gd_like_weights / gd_select_weights): while selected, `w` points at like_w and the
    // sample weights wait in w_main
    double* like_w = nullptr;
    double* w_main = nullptr;
    bool w_main_integral = false;
    // total of the weights `w` points at, added in a fixed order at upload; 0 = not known (auxiliary weights)
    double w_sum = 0, w_main_sum = 0;
    int w_sel = 0;
    bool borrowed = false;  // cols / w belong to another context of this process (gd_attach_samples)
    long long* wcum = nullptr;  // inclusive cumulative integer sample weights (thin.hip), built on first use
    std::shared_ptr<BucketCols> bq;  // bucket columns of the resident sample set (nullptr until the first upload)
    int64_t N = 0, n = 0, ld = 0;
    // reusable scratch (grown on demand)
    void* scratch = nullptr;
    int64_t scratch_bytes = 0;
    void* scratch2 = nullptr;
    int64_t scratch2_bytes = 0;
    void* gather_index = nullptr;  // the index list of gd_gather_items (its destination may be one of the scratch blocks)
    int64_t gather_index_bytes = 0;
    FftPlanCache* fft = nullptr;
    std::map<int, double*> dctmat;  // F -> F x F DCT-II matrix 2cos(pi k (2n+1) / 2F) (kopt2d.hip)
    std::map<int, void*> fft_tw;    // S -> the S twiddles e^{-2 pi i k / S} of the LDS transforms (density2d.hip)
    // page-locked staging ring for the small per-call tables of the entry points that return before their kernels
    // have run (gd_stage_h2d): a slot is reused only after the copy that read it has executed
    struct StageSlot {
        void* host = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool used = false;
        bool own = false;  // host is a private allocation (table larger than kStageBytes), not a slice of stage_block
    };
    static constexpr size_t kStageBytes = 64u << 10;
    void* stage_block = nullptr;
    // marks on the copy stream (gd_copy_mark / gd_copy_wait): a caller waits for ITS result copies only
    static constexpr int kCopyMarks = 16;
    hipEvent_t copy_marks[kCopyMarks] = {};
    int copy_mark_next = 0;
    static constexpr int kStageSlots = 32;
    StageSlot stage[kStageSlots];
    int stage_next = 0;
    // page-locked bounce block for small results (gd_fetch / gd_stream_sync)
    struct Fetch {
        void* dst;
        size_t off, bytes;
    };
    static constexpr size_t kFetchBytes = 4u << 20, kFetchMax = 1u << 20;
    void* fetch_block = nullptr;
    size_t fetch_off = 0;
    std::vector<Fetch> fetch_pending;
    // gd_kopt2d_enqueue / gd_kopt2d_finish: "stage A has run" events a second context's stream waits for
    static constexpr int kKoptEvents = 32;
    hipEvent_t kopt_evs[kKoptEvents] = {};
    int kopt_ev_next = 0;
    // RCCL communicator of a multi-GPU job (comm.hip) and its staging block
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    void* comm_buf = nullptr;
    size_t comm_buf_bytes = 0;
    // gd_comm_abandon (the ONE entry another thread may call on a context that is inside gd_comm_*): a set-up call that
    // comes back after the caller gave up on it must not install a communicator
    std::atomic<int> comm_abandon_gen{0};
    int comm_completed = 0;  // host-vector collectives that have completed since gd_comm_init (the first one is set-up)
    // gd_density2d_batch (batch2d.hip): device-block pool, cached index columns and the last call's blocks in flight
    void* batch_state = nullptr;
    void (*batch_state_release)(gd_ctx*, bool destroy) = nullptr;
};

// Stream-ordered H2D copy of a small host table whose storage the caller may release as soon as this returns.
int gd_stage_h2d(gd_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int gd_h2d(gd_ctx* ctx, void* d_dst, const void* src, size_t bytes);  // the same for any size (large tables by DMA)

// fft.hip: batched 2D real FFTs through rocFFT (plans cached per ctx); n0 = slow axis, n1 = fast axis.
// r2c: in batch x n0 x n1 doubles -> out batch x n0 x (n1/2+1) complex.  c2r is unnormalised and
// overwrites its input.
int gd_fft_r2c_2d(gd_ctx* ctx, int n0, int n1, int batch, const double* d_in, double2* d_out);
int gd_fft_c2r_2d(gd_ctx* ctx, int n0, int n1, int batch, double2* d_in, double* d_out);
void gd_fft_cache_destroy(gd_ctx* ctx);

// Small results come back through a page-locked bounce block that a copy KERNEL fills: a pageable destination turns
// hipMemcpyAsync into a blocking staged copy per call, and any DMA copy queues behind the 100-MB result copies of the
// previous batched call on the copy engines (measured: 0.1-2 ms per small copy while those are in flight).  gd_fetch
// enqueues the copy on ctx->stream; gd_stream_sync waits for the stream and delivers every pending fetch to its
// destination.  Results above kFetchMax (or when the block is full) take the DMA path.  gd_fetch_pinned writes straight
// into a page-locked destination (no delivery step: for entry points that return before their kernels have run).
int gd_fetch(gd_ctx* ctx, void* host_dst, const void* d_src, size_t bytes);
int gd_fetch_pinned(gd_ctx* ctx, void* pinned_dst, const void* d_src, size_t bytes);
int gd_stream_sync(gd_ctx* ctx);

int gd_fail(gd_ctx* ctx, int code, const char* fmt, ...);
void gd_comm_release(gd_ctx* ctx);  // comm.hip
bool gd_ctx_alive(gd_ctx* ctx);
int gd_stream_priority(gd_ctx* ctx, int level);  // re-create the (idle) compute stream: > 0 most urgent, 0 default, < 0 least urgent  // false once gd_destroy has run on it (core.hip)
void* gd_scratch(gd_ctx* ctx, int64_t bytes);   // returns nullptr (and sets err) on failure
void* gd_scratch2(gd_ctx* ctx, int64_t bytes);

#define GD_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return gd_fail(ctx, e__ == hipErrorOutOfMemory ? GD_ERR_NOMEM : GD_ERR_HIP, "%s: %s (%s:%d)", #call, \
                           hipGetErrorString(e__), __FILE__, __LINE__);                              \
    } while (0)

#define GD_KERNEL_CHECK() GD_HIP(hipGetLastError())

#define GD_TRY(call)              \
    do {                          \
        const int rc__ = (call);  \
        if (rc__) return rc__;    \
    } while (0)

#define GD_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) return gd_fail(ctx, GD_ERR_BADARG, "%s", msg); \
    } while (0)

// ---- device helpers -------------------------------------------------------------------------------
#define WAVE 64

// A pointer that a kernel reads out of a per-pair table in memory is "generic" to the compiler: it emits flat_load,
// which also counts on lgkmcnt, so the wait for the loaded data drains every LDS atomic the wave has in flight.
// Every such pointer here is device memory: say so, and the loads become global_load (vmcnt only).
typedef double gd_f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int gd_u32x4 __attribute__((ext_vector_type(4)));
// 16-byte loads from device memory (p 16-byte aligned)
__device__ __forceinline__ double2 gload_d2(const double* p) {
    const gd_f64x2 v = *(const __attribute__((address_space(1))) gd_f64x2*)p;
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ uint4 gload_u4(const void* p) {
    const gd_u32x4 v = *(const __attribute__((address_space(1))) gd_u32x4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, WAVE));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, WAVE));
    return v;
}

// Block-wide sum; result valid in thread 0.  `red` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ double block_min(double v, double* red) {
    v = wave_min(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = red[0];
    if (threadIdx.x == 0)
        for (int i = 1; i < nw; ++i) r = fmin(r, red[i]);
    return r;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = red[0];
    if (threadIdx.x == 0)
        for (int i = 1; i < nw; ++i) r = fmax(r, red[i]);
    return r;
}
