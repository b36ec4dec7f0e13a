// rocFFT plumbing: cached batched 2D real<->hermitian plans on the ctx stream.
// rocFFT is used for the convolution step (and the 256^2 power spectrum of the 2D bandwidth optimiser)
// only; everything around it is hand-written HIP.
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
    const size_t lengths[2] = {(size_t)n1, (size_t)n0};  // rocFFT: fastest dimension first
    rocfft_status st = rocfft_plan_create(&p.plan, rocfft_placement_notinplace,
                                          forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                          rocfft_precision_double, 2, lengths, (size_t)batch, nullptr);
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

int gd_fft_r2c_2d(gd_ctx* ctx, int n0, int n1, int batch, const double* d_in, double2* d_out) {
    FftPlan* p;
    int rc = get_plan(ctx, true, n0, n1, batch, &p);
    if (rc) return rc;
    void* in[1] = {(void*)d_in};
    void* out[1] = {(void*)d_out};
    rocfft_status st = rocfft_execute(p->plan, in, out, p->info);
    if (st != rocfft_status_success) return gd_fail(ctx, GD_ERR_FFT, "rocfft_execute(r2c) failed: %d", (int)st);
    return GD_OK;
}

int gd_fft_c2r_2d(gd_ctx* ctx, int n0, int n1, int batch, double2* d_in, double* d_out) {
    FftPlan* p;
    int rc = get_plan(ctx, false, n0, n1, batch, &p);
    if (rc) return rc;
    void* in[1] = {(void*)d_in};
    void* out[1] = {(void*)d_out};
    rocfft_status st = rocfft_execute(p->plan, in, out, p->info);
    if (st != rocfft_status_success) return gd_fail(ctx, GD_ERR_FFT, "rocfft_execute(c2r) failed: %d", (int)st);
    return GD_OK;
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
