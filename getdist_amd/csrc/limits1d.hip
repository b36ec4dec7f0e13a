// Equal-density credible limits of a batch of 1D densities (densities.py:186-248 initLimitGrids + getLimits):
// spline-refine each density to ~20000 points, find the density level below which a fraction (1 - contour) of the
// refined mass lies, and locate where the refined density crosses that level from either side.
//
// The reference does this per parameter on the host (splrep + splev of 19438 points + np.sort + np.cumsum: ~2 ms);
// C5 asks for the marginalised limits of 200 parameters at once.  One 1024-thread block per parameter:
//   1. not-a-knot cubic spline in second-derivative form on the uniform grid (the interpolant splrep(s=0) builds;
//      same formulation as getdist_amd/densities.py::NotAKnotSpline): M_1 and M_{n-2} are the plain second
//      differences, the interior is a (1,4,1) tridiagonal solve (Thomas, one thread, coefficients in LDS);
//   2. refined grid to global scratch (stays in L2), its sum by block reduction;
//   3. per contour an MSB radix select (8 passes x 8 key bits, LDS histograms of mass) for the ranked value at the
//      crossing, the tie count there, the mass strictly below and the next ranked value -- the four numbers the
//      reference reads off its sorted array and cumulative sum;
//   4. first / last refined index above the level by block min / max reduction, linear interpolation to the crossing.
#include "ctx.hpp"

#define LM_MAXC 8
#define LM_T 1024

__device__ __forceinline__ unsigned long long lm_key(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double lm_unkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

struct LimArgs {
    int F, factor, nc, bign;
};

__global__ void __launch_bounds__(LM_T) k_limits1d(const double* __restrict__ Pall, const double* __restrict__ x0s,
                                                   const double* __restrict__ spacings, const double* __restrict__ contours,
                                                   LimArgs A, double* __restrict__ Gall, double* __restrict__ out,
                                                   int* __restrict__ status) {
    extern __shared__ double sh[];
    const int F = A.F, tid = threadIdx.x, b = blockIdx.x;
    double* y = sh;        // F
    double* M = y + F;     // F second derivatives (index units)
    double* cp = M + F;    // F Thomas coefficients
    __shared__ double hist[256];
    __shared__ double red[16];
    __shared__ unsigned long long s_prefix;
    __shared__ double s_cum, s_bc;
    __shared__ int s_int[LM_T / 64];
    __shared__ unsigned long long s_key[LM_T / 64];
    const double* P = Pall + (int64_t)b * F;
    double* G = Gall + (int64_t)b * A.bign;
    for (int i = tid; i < F; i += LM_T) y[i] = P[i];
    __syncthreads();
    // ---- spline second derivatives
    for (int i = tid; i < F; i += LM_T) M[i] = (i >= 1 && i <= F - 2) ? 6.0 * ((y[i + 1] - y[i]) - (y[i] - y[i - 1])) : 0.0;  // rhs
    __syncthreads();
    if (tid == 0) {
        const int n = F;
        M[1] = M[1] / 6.0;
        M[n - 2] = M[n - 2] / 6.0;
        // unknowns i = 2 .. n-3:  M[i-1] + 4 M[i] + M[i+1] = rhs[i], M[1] and M[n-2] known
        if (n >= 6) {
            double cprev = 0.0, dprev = 0.0;
            for (int i = 2; i <= n - 3; ++i) {
                double r = M[i];
                if (i == 2) r -= M[1];
                if (i == n - 3) r -= M[n - 2];
                const double den = 4.0 - ((i == 2) ? 0.0 : cprev);
                cprev = 1.0 / den;
                dprev = (r - ((i == 2) ? 0.0 : dprev)) / den;
                cp[i] = cprev;
                M[i] = dprev;
            }
            for (int i = n - 4; i >= 2; --i) M[i] = M[i] - cp[i] * M[i + 1];
        } else if (n == 5) {
            M[2] = (M[2] - M[1] - M[3]) / 4.0;
        }
        M[0] = 2.0 * M[1] - M[2];
        M[n - 1] = 2.0 * M[n - 2] - M[n - 3];
    }
    __syncthreads();
    // ---- refined grid
    const double invf = 1.0 / (double)A.factor;
    double s = 0;
    for (int j = tid; j < A.bign; j += LM_T) {
        int k = j / A.factor;
        if (k > F - 2) k = F - 2;
        const double t = (double)(j - k * A.factor) * invf;
        const double c1 = (y[k + 1] - y[k]) - (2.0 * M[k] + M[k + 1]) / 6.0;
        const double c2 = M[k] / 2.0;
        const double c3 = (M[k + 1] - M[k]) / 6.0;
        const double v = (j % A.factor == 0) ? y[j / A.factor] : y[k] + t * (c1 + t * (c2 + t * c3));
        G[j] = v;
        s += v;
    }
    s = block_sum(s, red);
    if (tid == 0) s_bc = s - 0.5 * y[F - 1] - 0.5 * y[0];
    __syncthreads();
    const double norm = s_bc;
    const double x0 = x0s[b], fine = spacings[b] / (double)A.factor;
    int st = GD_OK;
    for (int ci = 0; ci < A.nc; ++ci) {
        const double target = (1.0 - contours[ci]) * norm;
        if (tid == 0) s_prefix = 0ull, s_cum = 0.0;
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0.0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            for (int j = tid; j < A.bign; j += LM_T) {
                const double v = G[j];
                const unsigned long long key = lm_key(v);
                if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], v);
            }
            __syncthreads();
            if (tid == 0) {
                double cum = s_cum;
                int pick = -1, last = -1;
                double cum_last = cum;
                for (int q = 0; q < 256; ++q) {
                    const double hv = hist[q];
                    if (hv != 0) {
                        last = q;
                        cum_last = cum;
                        if (cum + hv >= target) {
                            pick = q;
                            break;
                        }
                    }
                    cum += hv;
                }
                if (pick < 0) {
                    pick = last < 0 ? 0 : last;
                    cum = cum_last;
                }
                s_prefix = (prefix << 8) | (unsigned long long)pick;
                s_cum = cum;
            }
            __syncthreads();
        }
        const unsigned long long sel = s_prefix;
        // ties at the selected value, whether anything ranks below it, and the next ranked key above it
        int nt = 0, below = 0;
        unsigned long long nk = ~0ull;
        for (int j = tid; j < A.bign; j += LM_T) {
            const unsigned long long key = lm_key(G[j]);
            nt += (key == sel);
            below |= (key < sel);
            if (key > sel && key < nk) nk = key;
        }
        for (int o = 32; o > 0; o >>= 1) {
            nt += __shfl_down(nt, o, WAVE);
            below |= __shfl_down(below, o, WAVE);
            const unsigned long long ok = __shfl_down(nk, o, WAVE);
            if (ok < nk) nk = ok;
        }
        if ((tid & 63) == 0) s_int[tid >> 6] = nt | (below << 30), s_key[tid >> 6] = nk;
        __syncthreads();
        if (tid == 0) {
            nt = 0, below = 0, nk = ~0ull;
            for (int wv = 0; wv < LM_T / 64; ++wv) {
                nt += s_int[wv] & 0x3fffffff;
                below |= s_int[wv] >> 30;
                if (s_key[wv] < nk) nk = s_key[wv];
            }
            const double v = lm_unkey(sel);
            double cum_prev = s_cum, cum = s_cum;  // running mass before / after the selected entry
            int at = nt - 1;
            for (int a = 0; a < nt; ++a) {
                cum_prev = cum;
                cum += v;
                if (cum >= target) {
                    at = a;
                    break;
                }
            }
            double level = v;
            if (below || at > 0) {
                const double next = (at < nt - 1) ? v : ((nk == ~0ull) ? NAN : lm_unkey(nk));
                const double frac = (cum - target) / (cum - cum_prev);
                level = (1.0 - frac) * v + frac * next;
            }
            s_bc = level;
        }
        __syncthreads();
        const double level = s_bc;
        int imin = A.bign, imax = -1;
        for (int j = tid; j < A.bign; j += LM_T)
            if (G[j] > level) {
                if (j < imin) imin = j;
                if (j > imax) imax = j;
            }
        for (int o = 32; o > 0; o >>= 1) {
            imin = min(imin, __shfl_down(imin, o, WAVE));
            imax = max(imax, __shfl_down(imax, o, WAVE));
        }
        __syncthreads();
        if ((tid & 63) == 0) s_int[tid >> 6] = imin, s_key[tid >> 6] = (unsigned long long)(long long)imax;
        __syncthreads();
        if (tid == 0) {
            for (int wv = 0; wv < LM_T / 64; ++wv) {
                imin = min(imin, s_int[wv]);
                imax = max(imax, (int)(long long)s_key[wv]);
            }
            double* o = out + ((int64_t)b * A.nc + ci) * 4;
            const bool lim_bot = G[0] >= level, lim_top = G[A.bign - 1] >= level;
            double mn = x0, mx = x0 + (double)(F - 1) * spacings[b];
            if (!(level == level)) st = GD_ERR_SOLVER;
            if (!lim_bot) {
                if (imin >= A.bign || imin == 0) {
                    st = GD_ERR_SOLVER;
                } else {
                    const double d = (G[imin] - level) / (G[imin] - G[imin - 1]);
                    mn = x0 + ((double)imin - d) * fine;
                }
            }
            if (!lim_top) {
                if (imax < 0 || imax >= A.bign - 1) {
                    st = GD_ERR_SOLVER;
                } else {
                    const double d = (G[imax] - level) / (G[imax] - G[imax + 1]);
                    mx = x0 + ((double)imax + d) * fine;
                }
            }
            o[0] = mn, o[1] = mx, o[2] = lim_bot ? 1.0 : 0.0, o[3] = lim_top ? 1.0 : 0.0;
        }
        __syncthreads();
    }
    if (tid == 0) status[b] = st;
}

extern "C" {

int gd_limits1d(gd_ctx* ctx, int32_t B, int32_t F, const double* P, const double* x0, const double* spacing,
                const double* contours, int32_t nc, int32_t factor, double* out, int32_t* status_out) {
    GD_REQUIRE(ctx && P && x0 && spacing && contours && out && status_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096, "grid size out of range (8..4096)");
    GD_REQUIRE(nc >= 1 && nc <= LM_MAXC, "1..8 contours per call");
    if (factor <= 0) factor = 20000 / F > 2 ? 20000 / F : 2;  // densities.py:191-194
    GD_REQUIRE((int64_t)(F - 1) * factor + 1 <= (1 << 22), "refinement factor too large");
    LimArgs A{F, factor, nc, (F - 1) * factor + 1};
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_P = take((int64_t)B * F * 8), o_x = take((int64_t)B * 8), o_s = take((int64_t)B * 8), o_c = take(nc * 8),
                  o_G = take((int64_t)B * A.bign * 8), o_out = take((int64_t)B * nc * 32), o_st = take((int64_t)B * 4);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    GD_TRY(gd_h2d(ctx, base + o_P, P, (size_t)B * F * 8));
    GD_TRY(gd_h2d(ctx, base + o_x, x0, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, base + o_s, spacing, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, base + o_c, contours, (size_t)nc * 8));
    GD_HIP(hipFuncSetAttribute((const void*)k_limits1d, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4096 * 8));
    k_limits1d<<<B, LM_T, (size_t)3 * F * 8, ctx->stream>>>((const double*)(base + o_P), (const double*)(base + o_x),
                                                           (const double*)(base + o_s), (const double*)(base + o_c), A,
                                                           (double*)(base + o_G), (double*)(base + o_out), (int*)(base + o_st));
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, out, base + o_out, (size_t)B * nc * 32));
    GD_TRY(gd_fetch(ctx, status_out, base + o_st, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

}  // extern "C"
