// rocFFT plumbing: cached batched 2D real<->hermitian plans on the ctx stream.
// rocFFT serves the 256^2 power spectrum of the 2D bandwidth optimiser, the public getdist.convolve functions
// (convolve.hip) and the frame route of the 2D convolution (frames above 512, explicit masks, periodic axes); the
// batched triangle's convolutions run through the LDS transforms of density2d.hip.
#include <rocfft/rocfft.h>

#include <mutex>
#include <tuple>

#include "ctx.hpp"

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
};

struct FftPlanCache {
    std::map<std::tuple<int, int, int, int>, FftPlan> plans;  // (forward?, n0, n1, batch)
};

static std::once_flag g_rocfft_once;

static int get_plan(gd_ctx* ctx, bool forward, int n0, int n1, int batch, FftPlan** out) {
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    if (!ctx->fft) ctx->fft = new FftPlanCache();
    auto key = std::make_tuple(forward ? 1 : 0, n0, n1, batch);
    auto it = ctx->fft->plans.find(key);
    if (it != ctx->fft->plans.end()) {
        *out = &it->second;
        return GD_OK;
    }
    FftPlan p;
    const size_t lengths[2] = {(size_t)n1, (size_t)n0};  // rocFFT: fastest dimension first; n0 == 1 is a 1-D transform
    rocfft_status st = rocfft_plan_create(&p.plan, rocfft_placement_notinplace,
                                          forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                          rocfft_precision_double, n0 == 1 ? 1 : 2, lengths, (size_t)batch, nullptr);
    if (st != rocfft_status_success) return gd_fail(ctx, GD_ERR_FFT, "rocfft_plan_create(%dx%d x%d) failed: %d", n0, n1, batch, (int)st);
    st = rocfft_execution_info_create(&p.info);
    if (st != rocfft_status_success) return gd_fail(ctx, GD_ERR_FFT, "rocfft_execution_info_create failed: %d", (int)st);
    rocfft_plan_get_work_buffer_size(p.plan, &p.work_bytes);
    if (p.work_bytes) {
        if (hipMalloc(&p.work, p.work_bytes) != hipSuccess)
            return gd_fail(ctx, GD_ERR_NOMEM, "rocFFT work buffer of %zu bytes", p.work_bytes);
        rocfft_execution_info_set_work_buffer(p.info, p.work, p.work_bytes);
    }
    rocfft_execution_info_set_stream(p.info, ctx->stream);
    auto ins = ctx->fft->plans.emplace(key, p);
    *out = &ins.first->second;
    return GD_OK;
}

static int exec_plan(gd_ctx* ctx, bool forward, int n0, int n1, int batch, void* in, void* out) {
    FftPlan* p;
    int rc = get_plan(ctx, forward, n0, n1, batch, &p);
    if (rc) return rc;
    void* ib[1] = {in};
    void* ob[1] = {out};
    rocfft_execution_info_set_stream(p->info, ctx->stream);  // (the context's stream may have been re-created: gd_stream_priority)
    rocfft_status st = rocfft_execute(p->plan, ib, ob, p->info);
    if (st != rocfft_status_success)
        return gd_fail(ctx, GD_ERR_FFT, "rocfft_execute(%s %dx%d x%d) failed: %d", forward ? "r2c" : "c2r", n0, n1, batch, (int)st);
    return GD_OK;
}

// A rocFFT plan bakes in its batch count and costs 10-100 ms to build.  Small batches (interactive per-pair calls)
// are therefore executed as power-of-two sub-batches so that only {1,2,4,...,32} plans exist per frame size; large
// batches (the triangle) use one exact plan, amortised over the job.
static int exec_batched(gd_ctx* ctx, bool forward, int n0, int n1, int batch, char* in, char* out) {
    const size_t real_bytes = (size_t)n0 * n1 * 8, cplx_bytes = (size_t)n0 * (n1 / 2 + 1) * 16;
    const size_t in_stride = forward ? real_bytes : cplx_bytes, out_stride = forward ? cplx_bytes : real_bytes;
    if (batch >= 64) return exec_plan(ctx, forward, n0, n1, batch, in, out);
    int done = 0;
    for (int chunk = 32; chunk >= 1; chunk >>= 1)
        while (batch - done >= chunk) {
            int rc = exec_plan(ctx, forward, n0, n1, chunk, in + (size_t)done * in_stride, out + (size_t)done * out_stride);
            if (rc) return rc;
            done += chunk;
        }
    return GD_OK;
}

int gd_fft_r2c_2d(gd_ctx* ctx, int n0, int n1, int batch, const double* d_in, double2* d_out) {
    return exec_batched(ctx, true, n0, n1, batch, (char*)d_in, (char*)d_out);
}

int gd_fft_c2r_2d(gd_ctx* ctx, int n0, int n1, int batch, double2* d_in, double* d_out) {
    return exec_batched(ctx, false, n0, n1, batch, (char*)d_in, (char*)d_out);
}

void gd_fft_cache_destroy(gd_ctx* ctx) {
    if (!ctx->fft) return;
    for (auto& kv : ctx->fft->plans) {
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
        if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
        if (kv.second.work) (void)hipFree(kv.second.work);
    }
    delete ctx->fft;
    ctx->fft = nullptr;
}
