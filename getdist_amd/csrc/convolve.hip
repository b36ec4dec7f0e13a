// Stand-alone convolution entry points behind getdist_amd/convolve.py (getdist/convolve.py:196-212, 326-478) and the
// likelihood statistics of MCSamples._setLikeStats (mcsamples.py:2216-2243).
//   gd_circ_convolve      out = irfft(rfft(a) * rfft(b)) on frames of equal size (1-D: n0 == 1), rocFFT
//   gd_convolve1d_direct  the direct sum np.convolve(x, y, "full") evaluates for short operands (convolve.py:201-202)
//   gd_autoconvolve       autoConvolve (convolve.py:458-478): power spectrum of the zero-padded vector transformed
//                         back; idct type I of a real sequence of length s/2 + 1 is the real inverse FFT of length s
//   gd_like_stats         one min / moment pass and one exponential pass over the loglikes column
#include "ctx.hpp"

__global__ void k_cmul_scale(double2* __restrict__ a, const double2* __restrict__ b, int64_t n, double scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 u = a[i], v = b[i];
        a[i] = make_double2((u.x * v.x - u.y * v.y) * scale, (u.x * v.y + u.y * v.x) * scale);
    }
}

__global__ void k_power_inplace(double2* __restrict__ z, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 u = z[i];
        z[i] = make_double2(u.x * u.x + u.y * u.y, 0.0);
    }
}

// frame[i] = (x[i] - mean) * w[i] for i < N (w == nullptr: unit), 0 up to s
__global__ void k_fill_centered(const double* __restrict__ x, const double* __restrict__ w, int64_t N, double mean, int64_t s,
                                double* __restrict__ frame) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s; i += (int64_t)gridDim.x * blockDim.x)
        frame[i] = i < N ? (x[i] - mean) * (w ? w[i] : 1.0) : 0.0;
}

// out[k] = frame[k] / s [/ (N - k)]
__global__ void k_autoconv_out(const double* __restrict__ frame, int64_t n, double s, int64_t N, int normalize,
                               double* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        double v = frame[k] / s;
        if (normalize) v /= (double)(N - k);
        out[k] = v;
    }
}

__global__ void k_conv1d_direct(const double* __restrict__ x, int64_t nx, const double* __restrict__ y, int64_t ny,
                                double* __restrict__ out) {
    const int64_t nout = nx + ny - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j0 = i - (ny - 1) > 0 ? i - (ny - 1) : 0, j1 = i < nx - 1 ? i : nx - 1;
        double s = 0.0;
        for (int64_t j = j0; j <= j1; ++j) s += x[j] * y[i - j];
        out[i] = s;
    }
}

// pass 1: min of L (and the first row attaining it), max, sum w, sum w L, sum w L^2;  pass 2: sum w exp(L - Lmin),
// sum w exp(-(L - Lmin)).  One block of partials per launch block; the host adds them up.
__global__ void __launch_bounds__(256) k_like_pass1(const double* __restrict__ L, const double* __restrict__ w, int64_t N,
                                                    double* __restrict__ part) {
    __shared__ double red[16];
    double mn = INFINITY, mx = -INFINITY, sw = 0, swl = 0, swl2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double v = L[i], wt = w ? w[i] : 1.0;
        mn = fmin(mn, v), mx = fmax(mx, v);
        sw += wt, swl += wt * v, swl2 += wt * v * v;
    }
    const double r0 = block_min(mn, red), r1 = block_max(mx, red), r2 = block_sum(sw, red), r3 = block_sum(swl, red),
                 r4 = block_sum(swl2, red);
    if (threadIdx.x == 0) {
        double* p = part + (int64_t)blockIdx.x * 5;
        p[0] = r0, p[1] = r1, p[2] = r2, p[3] = r3, p[4] = r4;
    }
}

__global__ void __launch_bounds__(256) k_like_pass2(const double* __restrict__ L, const double* __restrict__ w, int64_t N,
                                                    double lmin, double* __restrict__ part,
                                                    unsigned long long* __restrict__ first_min) {
    __shared__ double red[16];
    double sp = 0, sm = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double v = L[i], wt = w ? w[i] : 1.0, d = v - lmin;
        sp += wt * exp(d), sm += wt * exp(-d);
        if (v == lmin) atomicMin(first_min, (unsigned long long)i);
    }
    const double r0 = block_sum(sp, red), r1 = block_sum(sm, red);
    if (threadIdx.x == 0) {
        double* p = part + (int64_t)blockIdx.x * 2;
        p[0] = r0, p[1] = r1;
    }
}

int gd_fft_r2c_2d(gd_ctx* ctx, int n0, int n1, int batch, const double* d_in, double2* d_out);
int gd_fft_c2r_2d(gd_ctx* ctx, int n0, int n1, int batch, double2* d_in, double* d_out);

extern "C" {

int gd_circ_convolve(gd_ctx* ctx, int32_t n0, int32_t n1, const double* a, const double* b, double* out) {
    GD_REQUIRE(ctx && a && b && out && n0 >= 1 && n1 >= 2, "bad argument");
    const int64_t nr = (int64_t)n0 * n1, nc = (int64_t)n0 * (n1 / 2 + 1);
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t o_a = 0, o_b = up(nr * 8), o_za = o_b + up(nr * 8), o_zb = o_za + up(nc * 16), total = o_zb + up(nc * 16);
    char* base = (char*)gd_scratch(ctx, total);
    if (!base) return GD_ERR_NOMEM;
    double *d_a = (double*)(base + o_a), *d_b = (double*)(base + o_b);
    double2 *za = (double2*)(base + o_za), *zb = (double2*)(base + o_zb);
    GD_TRY(gd_h2d(ctx, d_a, a, (size_t)nr * 8));
    GD_TRY(gd_h2d(ctx, d_b, b, (size_t)nr * 8));
    int rc;
    if ((rc = gd_fft_r2c_2d(ctx, n0, n1, 1, d_a, za))) return rc;
    if ((rc = gd_fft_r2c_2d(ctx, n0, n1, 1, d_b, zb))) return rc;
    k_cmul_scale<<<(unsigned)((nc + 255) / 256 > 4096 ? 4096 : (nc + 255) / 256), 256, 0, ctx->stream>>>(za, zb, nc, 1.0 / (double)nr);
    GD_KERNEL_CHECK();
    if ((rc = gd_fft_c2r_2d(ctx, n0, n1, 1, za, d_a))) return rc;
    GD_TRY(gd_fetch(ctx, out, d_a, (size_t)nr * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_convolve1d_direct(gd_ctx* ctx, const double* x, int64_t nx, const double* y, int64_t ny, double* out_full) {
    GD_REQUIRE(ctx && x && y && out_full && nx >= 1 && ny >= 1, "bad argument");
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t nout = nx + ny - 1, o_y = up(nx * 8), o_o = o_y + up(ny * 8);
    char* base = (char*)gd_scratch(ctx, o_o + up(nout * 8));
    if (!base) return GD_ERR_NOMEM;
    double *d_x = (double*)base, *d_y = (double*)(base + o_y), *d_o = (double*)(base + o_o);
    GD_TRY(gd_h2d(ctx, d_x, x, (size_t)nx * 8));
    GD_TRY(gd_h2d(ctx, d_y, y, (size_t)ny * 8));
    k_conv1d_direct<<<(unsigned)((nout + 255) / 256 > 4096 ? 4096 : (nout + 255) / 256), 256, 0, ctx->stream>>>(d_x, nx, d_y, ny, d_o);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, out_full, d_o, (size_t)nout * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_autoconvolve(gd_ctx* ctx, int32_t col, double mean, int32_t use_weights, const double* x_host, int64_t nx, int64_t s,
                    int64_t n, int32_t normalize, double* out) {
    GD_REQUIRE(ctx && out && s >= 2 && (s % 2) == 0 && n >= 1, "bad argument");
    int64_t N = nx;
    if (!x_host) {
        GD_REQUIRE(ctx->cols && col >= 0 && col < ctx->n + GD_EXTRA_COLS, "column out of range");
        N = ctx->N;
    }
    GD_REQUIRE(N >= 1 && s >= N && n <= N && s <= 0x7fffffff, "bad sizes");
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t nc = s / 2 + 1, o_z = up(s * 8), o_o = o_z + up(nc * 16);
    char* base = (char*)gd_scratch(ctx, o_o + up(n * 8));
    if (!base) return GD_ERR_NOMEM;
    double* frame = (double*)base;
    double2* z = (double2*)(base + o_z);
    double* d_out = (double*)(base + o_o);
    if (x_host) {
        GD_TRY(gd_h2d(ctx, frame, x_host, (size_t)N * 8));
        GD_HIP(hipMemsetAsync(frame + N, 0, (size_t)(s - N) * 8, ctx->stream));
    } else {
        k_fill_centered<<<4096, 256, 0, ctx->stream>>>(ctx->cols + (int64_t)col * ctx->ld, use_weights ? ctx->w : nullptr, N, mean, s,
                                                      frame);
        GD_KERNEL_CHECK();
    }
    int rc;
    if ((rc = gd_fft_r2c_2d(ctx, 1, (int)s, 1, frame, z))) return rc;
    k_power_inplace<<<4096, 256, 0, ctx->stream>>>(z, nc);
    GD_KERNEL_CHECK();
    if ((rc = gd_fft_c2r_2d(ctx, 1, (int)s, 1, z, frame))) return rc;
    k_autoconv_out<<<(unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0, ctx->stream>>>(frame, n, (double)s, N, normalize,
                                                                                                      d_out);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, out, d_out, (size_t)n * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_like_stats(gd_ctx* ctx, int32_t col, double* out8) {
    GD_REQUIRE(ctx && out8, "bad argument");
    GD_REQUIRE(ctx->cols && col >= 0 && col < ctx->n + GD_EXTRA_COLS, "column out of range");
    const int nblk = 1024;
    char* base = (char*)gd_scratch(ctx, (int64_t)nblk * 5 * 8 + 256);
    if (!base) return GD_ERR_NOMEM;
    double* d_part = (double*)base;
    unsigned long long* d_first = (unsigned long long*)(base + (int64_t)nblk * 5 * 8);
    const double* L = ctx->cols + (int64_t)col * ctx->ld;
    k_like_pass1<<<nblk, 256, 0, ctx->stream>>>(L, ctx->w, ctx->N, d_part);
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)nblk * 5);
    GD_TRY(gd_fetch(ctx, h.data(), d_part, h.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    double mn = INFINITY, mx = -INFINITY, sw = 0, swl = 0, swl2 = 0;
    for (int b = 0; b < nblk; ++b) {
        mn = fmin(mn, h[(size_t)b * 5]), mx = fmax(mx, h[(size_t)b * 5 + 1]);
        sw += h[(size_t)b * 5 + 2], swl += h[(size_t)b * 5 + 3], swl2 += h[(size_t)b * 5 + 4];
    }
    const unsigned long long none = ~0ull;
    GD_TRY(gd_h2d(ctx, d_first, &none, 8));
    k_like_pass2<<<nblk, 256, 0, ctx->stream>>>(L, ctx->w, ctx->N, mn, d_part, d_first);
    GD_KERNEL_CHECK();
    unsigned long long first = 0;
    GD_TRY(gd_fetch(ctx, h.data(), d_part, (size_t)nblk * 2 * 8));
    GD_TRY(gd_fetch(ctx, &first, d_first, 8));
    GD_TRY(gd_stream_sync(ctx));
    double sp = 0, sm = 0;
    for (int b = 0; b < nblk; ++b) sp += h[(size_t)b * 2], sm += h[(size_t)b * 2 + 1];
    out8[0] = mn, out8[1] = mx, out8[2] = sw, out8[3] = swl, out8[4] = swl2, out8[5] = sp, out8[6] = sm, out8[7] = (double)first;
    return GD_OK;
}

}  // extern "C"
