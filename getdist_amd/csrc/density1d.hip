// 1D density on the device: DCT-II for the ISJ bandwidth functional, Gaussian tap convolution,
// boundary correction (orders 0/1/2), multiplicative bias correction, max-normalisation.
// One workgroup per parameter; everything (F <= 4096 bins, <= F taps) lives in LDS.
#include "ctx.hpp"
#include "solvers.hpp"

// ---- DCT-II (scipy.fftpack.dct type 2, unnormalised): a[k] = 2 sum_n x[n] cos(pi k (2n+1) / (2F)) ----------
// tab[m] = cos(pi m / (2F)), m in [0,4F), built with exact octant reduction.
__global__ void k_cos_table(int F, double* __restrict__ tab) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= 4 * F) return;
    // angle = pi*m/(2F); reduce to [0, pi/2] exactly in integers
    int q = m / F;          // quadrant 0..3
    int r = m - q * F;      // angle = q*pi/2 + pi*r/(2F)
    const double c = cospi((double)r / (2.0 * F)), s = sinpi((double)r / (2.0 * F));
    double v;
    switch (q) {
        case 0: v = c; break;
        case 1: v = -s; break;
        case 2: v = -c; break;
        default: v = s; break;
    }
    tab[m] = v;
}

// grid (ceil(F/256), B); hist/out are device arrays B x F
__global__ void __launch_bounds__(256) k_dct1d(const double* __restrict__ hist, int F, const double* __restrict__ tab,
                                               double* __restrict__ out) {
    extern __shared__ double xs[];  // F normalised data
    __shared__ double red[16];
    const double* h = hist + (int64_t)blockIdx.y * F;
    double s = 0;
    for (int i = threadIdx.x; i < F; i += 256) s += h[i];
    s = block_sum(s, red);
    __shared__ double total;
    if (threadIdx.x == 0) total = s;
    __syncthreads();
    for (int i = threadIdx.x; i < F; i += 256) xs[i] = h[i] / total;
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= F) return;
    const int F4 = 4 * F;
    int m = k % F4, step = (2 * k) % F4;
    double a0 = 0, a1 = 0;
    int n = 0;
    for (; n + 1 < F; n += 2) {
        a0 = fma(xs[n], tab[m], a0);
        m += step;
        if (m >= F4) m -= F4;
        a1 = fma(xs[n + 1], tab[m], a1);
        m += step;
        if (m >= F4) m -= F4;
    }
    if (n < F) a0 = fma(xs[n], tab[m], a0);
    out[(int64_t)blockIdx.y * F + k] = 2.0 * (a0 + a1);
}

// ---- Botev improved-Sheather-Jones bandwidth, solved on the device (kde_bandwidth.py:59-73,102-135) ---------------
// One block per parameter.  The fixed-point functional is a chain of six 1023-term sums (block reductions over
// coefficients held in LDS); the root finder is MINPACK's hybrd for one unknown exactly as scipy's fsolve drives it
// (solvers.hpp, checked against scipy evaluation by evaluation), followed by the reference's brentq re-check.  All
// threads run the scalar control flow redundantly on identical values, so every branch is block-uniform.
struct IsjConsts {
    double two_pi_pow[8];  // 2 * pi^(2 l), l = 2..7
    double kde_const[8];   // _kde_consts_1d for j = 6..2, indexed by j
    double rootpi, pisq;
};

__device__ __forceinline__ double block_sum_all(double v, double* red, double* slot) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
        for (int i = 0; i < nw; ++i) r += red[i];
        *slot = r;
    }
    __syncthreads();
    return *slot;
}

// grid B; a_all = DCT-II coefficients (B x F); out[b] = {hfrac, status}
__global__ void __launch_bounds__(256) k_isj1d(const double* __restrict__ a_all, int F, const double* __restrict__ neff,
                                               const double* __restrict__ nscale, IsjConsts C, double* __restrict__ out) {
    extern __shared__ double sh[];
    double* a2 = sh;            // F-1
    double* logI = sh + F;      // F-1
    __shared__ double red[8];
    __shared__ double slot;
    const int b = blockIdx.x, tid = threadIdx.x, K = F - 1;
    const double* a = a_all + (int64_t)b * F;
    for (int k = tid; k < K; k += 256) {
        const double v = a[k + 1] / 2.0;
        a2[k] = v * v;
        const double I = (double)(k + 1) * (double)(k + 1);
        logI[k] = log(I);
    }
    __syncthreads();
    const double N = neff[b];
    auto functional = [&](int l, double t) {  // 2 pi^(2l) sum_k a2_k exp(l logI_k - I_k pi^2 t)
        const double pt = C.pisq * t;
        double s = 0;
        for (int k = tid; k < K; k += 256) {
            const double I = (double)(k + 1) * (double)(k + 1);
            s += a2[k] * exp((double)l * logI[k] - I * pt);
        }
        return C.two_pi_pow[l] * block_sum_all(s, red, &slot);
    };
    auto fixed_point = [&](double h, bool* fail) -> double {
        if (h <= 0) return h - 1;
        double f = functional(7, h * h);
        for (int j = 6; j >= 2; --j) {
            const double t_j = pow(C.kde_const[j] / N / f, 2 / (3.0 + 2 * j));
            f = functional(j, t_j);
            if (f == 0.0) {  // "zero f in _bandwidth_fixed_point (non-convergence)"
                *fail = true;
                return 0.0;
            }
        }
        return h - pow(2 * N * C.rootpi * f, -1.0 / 5);
    };
    const double n_scaling = nscale[b];
    double hfrac = 0.53 * n_scaling;
    const gdsolve::HybrdResult hr = gdsolve::hybrd1(fixed_point, hfrac, hfrac / 20, 400, 1.0);
    int status = GD_OK;
    if (hr.info < 0) {
        status = GD_ERR_SOLVER;  // an exception inside fsolve: the reference logs and returns None
    } else {
        hfrac = hr.x;
        if (hfrac < 0.019 * n_scaling) {  // may be the second solution: re-check with Brent (kde_bandwidth.py:124-131)
            const double xtol = hfrac / 20;
            if (xtol > 0) {
                const gdsolve::BrentResult br = gdsolve::brentq(fixed_point, 0.019 * n_scaling, 0.5, xtol,
                                                                4.0 * gdsolve::EPSMCH, 100);
                if (br.status == 0) hfrac = br.x;
            }
        }
    }
    if (tid == 0) {
        out[2 * b] = hfrac;
        out[2 * b + 1] = (double)status;
    }
}

// ---- density assembly -------------------------------------------------------------------------------------
struct D1Args {
    int F, bco, mbc;
};

__device__ __forceinline__ double edge_mask(int idx, int F, bool bot, bool top) {
    // prior_mask of mcsamples.py:1602-1608 expressed on the un-padded index idx = n - i
    if (idx < 0) return bot ? 0.0 : 1.0;
    if (idx == 0) return bot ? 0.5 : 1.0;
    if (idx > F - 1) return top ? 0.0 : 1.0;
    if (idx == F - 1) return top ? 0.5 : 1.0;
    return 1.0;
}

__global__ void __launch_bounds__(256) k_density1d(const double* __restrict__ hist, const double* __restrict__ smooth,
                                                   const int* __restrict__ winw, const int* __restrict__ flags,
                                                   D1Args A, double* __restrict__ Pout, int* __restrict__ status) {
    extern __shared__ double sh[];
    const int F = A.F, b = blockIdx.x, tid = threadIdx.x;
    double* bins = sh;          // F
    double* P = bins + F;       // F
    double* fine = P + F;       // F (also the circular copy in periodic mode)
    double* Win = fine + F;     // 2w+1 <= F
    __shared__ double red[16];
    __shared__ double bc;
    const int w = winw[b];
    const double hh = smooth[b];
    const bool bot = flags[b] & 1, top = flags[b] & 2, periodic = flags[b] & 4;
    const bool has_limits = bot || top;
    const int M = 2 * w + 1;
    for (int i = tid; i < F; i += 256) bins[i] = hist[(int64_t)b * F + i];
    // Kernel1D (mcsamples.py:129-135)
    double s = 0;
    for (int j = tid; j < M; j += 256) {
        const double x = (double)(j - w) / hh;
        const double v = exp(-(x * x) / 2.0);
        Win[j] = v;
        s += v;
    }
    s = block_sum(s, red);
    if (tid == 0) bc = s;
    __syncthreads();
    const double wsum = bc;
    for (int j = tid; j < M; j += 256) Win[j] = Win[j] / wsum;
    __syncthreads();
    const int Fc = F - 1;  // circular length in periodic mode (convolve.py:337-339)
    if (periodic) {
        for (int i = tid; i < Fc; i += 256) fine[i] = bins[i] + (i == 0 ? bins[F - 1] : 0.0);
        __syncthreads();
    }
    // first convolution + boundary correction, per output bin
    for (int n = tid; n < F; n += 256) {
        double p0 = 0;
        if (periodic) {
            const int nn = (n == F - 1) ? 0 : n;
            for (int i = -w; i <= w; ++i) {
                int idx = nn - i;
                idx %= Fc;
                if (idx < 0) idx += Fc;
                p0 = fma(Win[i + w], fine[idx], p0);
            }
            P[n] = p0;
            continue;
        }
        double xP = 0, x2P = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        const bool need_mask = has_limits && A.bco >= 0;
        for (int i = -w; i <= w; ++i) {
            const int idx = n - i;
            const double wi = Win[i + w];
            const double di = (double)i;
            const double xw = wi * di;        // kernel.Win * kernel.x
            const double x2w = xw * di;       // xWin * kernel.x
            const double v = (idx >= 0 && idx < F) ? bins[idx] : 0.0;
            p0 = fma(wi, v, p0);
            xP = fma(xw, v, xP);
            x2P = fma(x2w, v, x2P);
            if (need_mask) {
                const double m = edge_mask(idx, F, bot, top);
                a0 = fma(wi, m, a0);
                a1 = fma(xw, m, a1);
                a2 = fma(x2w, m, a2);
                a3 = fma(x2w * di, m, a3);
                a4 = fma(x2w * di * di, m, a4);
            }
        }
        double pv = p0;
        if (need_mask) {
            if (a0 * p0 != 0.0) {
                const double normed = p0 / a0;
                if (A.bco == 0) {
                    pv = normed;
                } else {
                    double corrected;
                    if (A.bco == 1) {
                        corrected = (p0 * a2 - xP * a1) / (a0 * a2 - a1 * a1);
                    } else {
                        const double denom = a4 * a2 * a0 - a4 * (a1 * a1) - a2 * a2 * a2 - a3 * a3 * a0 + 2 * a1 * a2 * a3;
                        const double Aq = a4 * a2 - a3 * a3, Bq = a2 * a3 - a4 * a1, Cq = a3 * a1 - a2 * a2;
                        corrected = (p0 * Aq + xP * Bq + x2P * Cq) / denom;
                    }
                    pv = normed * exp(fmin(corrected / normed, 4.0) - 1.0);
                }
            }
        } else if (A.bco == 2) {
            // higher-order kernel for unbounded parameters (mcsamples.py:1638-1647)
            double s2 = 0, s4 = 0;
            for (int i = -w; i <= w; ++i) {
                const double di = (double)i, xw2 = Win[i + w] * (di * di);
                s2 += xw2;
                s4 = fma(xw2, di * di, s4);
            }
            const double corrected = (p0 * s4 - s2 * x2P) / (s4 - s2 * s2);
            if (p0 > 0) pv = p0 * exp(fmin(corrected / p0, 2.0) - 1.0);
        }
        P[n] = pv;
    }
    __syncthreads();
    // multiplicative bias correction (mcsamples.py:1649-1666)
    for (int round = 0; round < A.mbc; ++round) {
        if (periodic) {
            // fine = bins / prob1, made circular
            for (int i = tid; i < Fc; i += 256) {
                const double p1 = (P[i] == 0.0) ? 1.0 : P[i];
                double v = bins[i] / p1;
                if (i == 0) {
                    const double pl = (P[F - 1] == 0.0) ? 1.0 : P[F - 1];
                    v += bins[F - 1] / pl;
                }
                fine[i] = v;
            }
        } else {
            for (int i = tid; i < F; i += 256) fine[i] = bins[i] / ((P[i] == 0.0) ? 1.0 : P[i]);
        }
        __syncthreads();
        double newp[16];  // F <= 4096 -> at most 16 outputs per thread
        int cnt = 0;
        for (int n = tid; n < F; n += 256, ++cnt) {
            double c0 = 0, a0 = 0;
            if (periodic) {
                const int nn = (n == F - 1) ? 0 : n;
                for (int i = -w; i <= w; ++i) {
                    int idx = (nn - i) % Fc;
                    if (idx < 0) idx += Fc;
                    c0 = fma(Win[i + w], fine[idx], c0);
                }
                newp[cnt] = P[n] * c0;
            } else {
                for (int i = -w; i <= w; ++i) {
                    const int idx = n - i;
                    if (idx >= 0 && idx < F) {
                        const double wi = Win[i + w];
                        c0 = fma(wi, fine[idx], c0);
                        const double m = (idx == 0 && bot) ? 0.5 : ((idx == F - 1 && top) ? 0.5 : 1.0);
                        a0 = fma(wi, m, a0);
                    }
                }
                newp[cnt] = (P[n] * c0) / a0;
            }
        }
        __syncthreads();
        cnt = 0;
        for (int n = tid; n < F; n += 256, ++cnt) P[n] = newp[cnt];
        __syncthreads();
    }
    double mx = -INFINITY;
    for (int n = tid; n < F; n += 256) mx = fmax(mx, P[n]);
    mx = block_max(mx, red);
    if (tid == 0) {
        bc = mx;
        status[b] = (mx == 0.0) ? GD_ERR_EMPTY : GD_OK;
    }
    __syncthreads();
    mx = bc;
    for (int n = tid; n < F; n += 256) Pout[(int64_t)b * F + n] = (mx == 0.0) ? 0.0 : P[n] / mx;
}

// ---- mean likelihoods (mcsamples.py:1672-1682): one block per parameter -------------------------------------
__global__ void __launch_bounds__(256) k_likes1d(const double* __restrict__ hist, const double* __restrict__ likehist,
                                                 const double* __restrict__ Pfinal, const double* __restrict__ smooth,
                                                 const int* __restrict__ winw, const int* __restrict__ flags, int F,
                                                 int shade_mean_loglikes, double* __restrict__ out, int* __restrict__ status) {
    extern __shared__ double sh[];
    const int b = blockIdx.x, tid = threadIdx.x;
    double* src = sh;        // F: operand of the current convolution (circular copy in periodic mode)
    double* raw = src + F;   // F: rawbins = conv(bins, Win)
    double* lk = raw + F;    // F: binlikes
    double* Win = lk + F;    // 2w+1 <= F
    __shared__ double red[16];
    __shared__ double bc;
    const int w = winw[b], M = 2 * w + 1, Fc = F - 1;
    const double hh = smooth[b];
    const bool periodic = flags[b] & 4;
    const double* bins = hist + (int64_t)b * F;
    const double* lh = likehist + (int64_t)b * F;
    const double* P = Pfinal + (int64_t)b * F;
    double s = 0;
    for (int j = tid; j < M; j += 256) {
        const double x = (double)(j - w) / hh;
        const double v = exp(-(x * x) / 2.0);
        Win[j] = v;
        s += v;
    }
    s = block_sum(s, red);
    if (tid == 0) bc = s;
    __syncthreads();
    const double wsum = bc;
    for (int j = tid; j < M; j += 256) Win[j] = Win[j] / wsum;
    // dst = convolve1D(v, Win, mode) for v given element-wise by `get`
    auto conv = [&](auto get, double* dst) {
        __syncthreads();
        if (periodic) {
            for (int i = tid; i < Fc; i += 256) src[i] = get(i) + (i == 0 ? get(F - 1) : 0.0);
        } else {
            for (int i = tid; i < F; i += 256) src[i] = get(i);
        }
        __syncthreads();
        for (int n = tid; n < F; n += 256) {
            double acc = 0;
            if (periodic) {
                const int nn = (n == F - 1) ? 0 : n;
                for (int i = -w; i <= w; ++i) {
                    int idx = (nn - i) % Fc;
                    if (idx < 0) idx += Fc;
                    acc = fma(Win[i + w], src[idx], acc);
                }
            } else {
                for (int i = -w; i <= w; ++i) {
                    const int idx = n - i;
                    if (idx >= 0 && idx < F) acc = fma(Win[i + w], src[idx], acc);
                }
            }
            dst[n] = acc;
        }
        __syncthreads();
    };
    conv([&](int i) { return bins[i]; }, raw);
    conv([&](int i) { return (P[i] > 0) ? lh[i] / P[i] : lh[i]; }, lk);
    for (int n = tid; n < F; n += 256)
        if (P[n] > 0) lk[n] = lk[n] * (P[n] / raw[n]);
    __syncthreads();
    if (shade_mean_loglikes) {
        double mn = INFINITY;
        for (int n = tid; n < F; n += 256) mn = fmin(mn, lk[n]);
        mn = block_min(mn, red);
        if (tid == 0) bc = mn;
        __syncthreads();
        mn = bc;
        for (int n = tid; n < F; n += 256) {
            const double d = lk[n] - mn;
            lk[n] = (raw[n] == 0.0) ? 0.0 : ((d < 30) ? exp(-d) : 0.0);
        }
        __syncthreads();
    }
    double mx = -INFINITY;
    for (int n = tid; n < F; n += 256) mx = fmax(mx, lk[n]);
    mx = block_max(mx, red);
    if (tid == 0) {
        bc = mx;
        status[b] = (mx == 0.0) ? GD_ERR_EMPTY : GD_OK;
    }
    __syncthreads();
    mx = bc;
    for (int n = tid; n < F; n += 256) out[(int64_t)b * F + n] = (mx == 0.0) ? 0.0 : lk[n] / mx;
}

extern "C" {

int gd_dct1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, double* a_out) {
    GD_REQUIRE(ctx && hist && a_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 2 && F <= 4096, "fine_bins out of range (2..4096)");
    const int64_t nb = (int64_t)B * F * 8;
    char* base = (char*)gd_scratch(ctx, 2 * nb + (int64_t)4 * F * 8 + 512);
    if (!base) return GD_ERR_NOMEM;
    double* d_in = (double*)base;
    double* d_out = (double*)(base + (nb + 255) / 256 * 256);
    double* d_tab = (double*)(base + 2 * ((nb + 255) / 256 * 256));
    GD_TRY(gd_h2d(ctx, d_in, hist, (size_t)nb));
    k_cos_table<<<(4 * F + 255) / 256, 256, 0, ctx->stream>>>(F, d_tab);
    GD_KERNEL_CHECK();
    k_dct1d<<<dim3((F + 255) / 256, B), 256, (size_t)F * 8, ctx->stream>>>(d_in, F, d_tab, d_out);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, a_out, d_out, (size_t)nb));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

// hist on the host (copied in) or, with `hist_on_device`, B x F doubles in device memory (read in place)
static int isj1d_core(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, bool hist_on_device, const double* neff,
                      double* hfrac_out, int32_t* status_out) {
    GD_REQUIRE(ctx && hist && neff && hfrac_out && status_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096, "fine_bins out of range (8..4096)");
    const int64_t nb = ((int64_t)B * F * 8 + 255) / 256 * 256, ns = ((int64_t)B * 16 + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, 2 * nb + 3 * ns + (int64_t)4 * F * 8 + 512);
    if (!base) return GD_ERR_NOMEM;
    const double* d_in = hist_on_device ? hist : (const double*)base;
    double* d_a = (double*)(base + nb);
    double* d_neff = (double*)(base + 2 * nb);
    double* d_nscale = (double*)(base + 2 * nb + ns);
    double* d_out = (double*)(base + 2 * nb + 2 * ns);
    double* d_tab = (double*)(base + 2 * nb + 3 * ns);
    // host-side constants with the reference's expressions (kde_bandwidth.py:47-56,63,69,73) in libm arithmetic
    IsjConsts C;
    const double pi = 3.141592653589793;
    C.rootpi = sqrt(pi);
    C.pisq = pi * pi;
    for (int l = 0; l < 8; ++l) {
        C.two_pi_pow[l] = 2 * pow(pi, (double)(2 * l));
        double prod = 1;
        for (int q = 1; q < 2 * l; q += 2) prod *= q;
        C.kde_const[l] = (1 + pow(0.5, l + 0.5)) / 3 * prod / (C.rootpi / sqrt(2.0));
    }
    std::vector<double> nscale((size_t)B);
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(neff[b] > 0, "effective sample number must be positive");
        nscale[b] = pow(neff[b], -1.0 / 5);
    }
    if (!hist_on_device) GD_TRY(gd_h2d(ctx, (double*)base, hist, (size_t)B * F * 8));
    GD_TRY(gd_h2d(ctx, d_neff, neff, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, d_nscale, nscale.data(), (size_t)B * 8));
    k_cos_table<<<(4 * F + 255) / 256, 256, 0, ctx->stream>>>(F, d_tab);
    GD_KERNEL_CHECK();
    k_dct1d<<<dim3((F + 255) / 256, B), 256, (size_t)F * 8, ctx->stream>>>(d_in, F, d_tab, d_a);
    GD_KERNEL_CHECK();
    GD_HIP(hipFuncSetAttribute((const void*)k_isj1d, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4096 * 8));
    k_isj1d<<<B, 256, (size_t)2 * F * 8, ctx->stream>>>(d_a, F, d_neff, d_nscale, C, d_out);
    GD_KERNEL_CHECK();
    std::vector<double> res((size_t)2 * B);
    GD_TRY(gd_fetch(ctx, res.data(), d_out, (size_t)B * 16));
    GD_TRY(gd_stream_sync(ctx));
    for (int b = 0; b < B; ++b) {
        hfrac_out[b] = res[2 * b];
        status_out[b] = (int32_t)res[2 * b + 1];
    }
    return GD_OK;
}

// hist, P_out on the host, or with `on_device` both in device memory (the density is left there: no copy, no sync)
static int density1d_core(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, bool hist_on_device, const double* smooth,
                          const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc, double* P_out, int32_t* status_out) {

    GD_REQUIRE(ctx && hist && smooth && winw && flags && P_out && status_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096, "fine_bins out of range (8..4096)");
    GD_REQUIRE(bco >= -1 && bco <= 2, "Unknown boundary_correction_order (expected 0, 1, 2)");
    GD_REQUIRE(mbc >= 0 && mbc <= 8, "mult_bias_correction_order out of range");
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(winw[b] >= 0 && 2 * winw[b] + 1 <= F, "window wider than the grid");
        GD_REQUIRE(smooth[b] > 0, "smoothing scale must be positive");
    }
    const int64_t nb = ((int64_t)B * F * 8 + 255) / 256 * 256, ns = ((int64_t)B * 8 + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, 2 * nb + 4 * ns);
    if (!base) return GD_ERR_NOMEM;
    const double* d_hist = hist_on_device ? hist : (const double*)base;
    double* d_P = (double*)(base + nb);
    double* d_smooth = (double*)(base + 2 * nb);
    int* d_winw = (int*)(base + 2 * nb + ns);
    int* d_flags = (int*)(base + 2 * nb + 2 * ns);
    int* d_status = (int*)(base + 2 * nb + 3 * ns);
    if (!hist_on_device) GD_TRY(gd_h2d(ctx, (double*)base, hist, (size_t)B * F * 8));
    GD_TRY(gd_h2d(ctx, d_smooth, smooth, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, d_winw, winw, (size_t)B * 4));
    GD_TRY(gd_h2d(ctx, d_flags, flags, (size_t)B * 4));
    D1Args A{F, bco, mbc};
    const size_t lds = (size_t)4 * F * 8;
    GD_HIP(hipFuncSetAttribute((const void*)k_density1d, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 8));
    k_density1d<<<B, 256, lds, ctx->stream>>>(d_hist, d_smooth, d_winw, d_flags, A, d_P, d_status);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, P_out, d_P, (size_t)B * F * 8));
    GD_TRY(gd_fetch(ctx, status_out, d_status, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_isj1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, const double* neff, double* hfrac_out,
             int32_t* status_out) {
    return isj1d_core(ctx, B, F, hist, false, neff, hfrac_out, status_out);
}
int gd_isj1d_dev(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* neff, double* hfrac_out,
                 int32_t* status_out) {
    return isj1d_core(ctx, B, F, (const double*)d_hist, true, neff, hfrac_out, status_out);
}
int gd_density1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, const double* smooth, const int32_t* winw,
                 const int32_t* flags, int32_t bco, int32_t mbc, double* P_out, int32_t* status_out) {
    return density1d_core(ctx, B, F, hist, false, smooth, winw, flags, bco, mbc, P_out, status_out);
}
int gd_density1d_dev(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* smooth, const int32_t* winw,
                     const int32_t* flags, int32_t bco, int32_t mbc, double* P_out, int32_t* status_out) {
    return density1d_core(ctx, B, F, (const double*)d_hist, true, smooth, winw, flags, bco, mbc, P_out, status_out);
}

int gd_likes1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, const double* likehist, const double* P,
               const double* smooth, const int32_t* winw, const int32_t* flags, int32_t shade_mean_loglikes,
               double* likes_out, int32_t* status_out) {
    GD_REQUIRE(ctx && hist && likehist && P && smooth && winw && flags && likes_out && status_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096, "fine_bins out of range (8..4096)");
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(winw[b] >= 0 && 2 * winw[b] + 1 <= F, "window wider than the grid");
        GD_REQUIRE(smooth[b] > 0, "smoothing scale must be positive");
    }
    const int64_t nb = ((int64_t)B * F * 8 + 255) / 256 * 256, ns = ((int64_t)B * 8 + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, 4 * nb + 4 * ns);
    if (!base) return GD_ERR_NOMEM;
    double *d_hist = (double*)base, *d_lh = (double*)(base + nb), *d_P = (double*)(base + 2 * nb),
           *d_out = (double*)(base + 3 * nb), *d_smooth = (double*)(base + 4 * nb);
    int *d_winw = (int*)(base + 4 * nb + ns), *d_flags = (int*)(base + 4 * nb + 2 * ns),
        *d_status = (int*)(base + 4 * nb + 3 * ns);
    GD_TRY(gd_h2d(ctx, d_hist, hist, (size_t)B * F * 8));
    GD_TRY(gd_h2d(ctx, d_lh, likehist, (size_t)B * F * 8));
    GD_TRY(gd_h2d(ctx, d_P, P, (size_t)B * F * 8));
    GD_TRY(gd_h2d(ctx, d_smooth, smooth, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, d_winw, winw, (size_t)B * 4));
    GD_TRY(gd_h2d(ctx, d_flags, flags, (size_t)B * 4));
    const size_t lds = (size_t)4 * F * 8;
    GD_HIP(hipFuncSetAttribute((const void*)k_likes1d, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 8));
    k_likes1d<<<B, 256, lds, ctx->stream>>>(d_hist, d_lh, d_P, d_smooth, d_winw, d_flags, F, shade_mean_loglikes, d_out,
                                            d_status);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, likes_out, d_out, (size_t)B * F * 8));
    GD_TRY(gd_fetch(ctx, status_out, d_status, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

}  // extern "C"
