// 2D bandwidth optimiser (Botev ISJ in two dimensions) on the device.
//
// Per pair: a2 = dct2d(hist/sum)[1:,1:]^2 by two small fp64 GEMMs against a cached DCT-II matrix, then ONE
// workgroup per pair runs the whole scalar solve without leaving the device: Brent's method on the
// fixed point t = xi(t) (scipy brentq control flow), each function evaluation being four dependent
// "levels" of bilinear forms  wy . a2 . wx  over the 255x255 coefficient grid (streamed from L2), then the
// psi functionals get_h needs, and -- when the pair is unbounded -- the odd functionals from |fft2|^2
// (rocFFT).  Reference: kde_bandwidth.py:146-270.
#include "ctx.hpp"
#include "ldsfft.hpp"
#include "solvers.hpp"

#ifndef KT
#define KT 512  // threads per pair: 512 leave 176 VGPRs and no spills (1024: 128 VGPRs, ~170 B spilled per lane; measured 5.8 -> 5.4 ms)
#endif
#define MAXF 6  // forms per level
#ifdef KOPT_PROFILE
__device__ long long g_prof[8];  // weights, tile loads, tile FMAs, reductions, times/pow, levels, evaluations
__device__ long long g_wave[8];  // k_kopt2d_res: cycles each wave of block 0 spent in the forms
#define PROF_T0 long long prof_t0 = clock64()
#define PROF_ADD(slot)                                                        \
    do {                                                                      \
        const long long prof_t1 = clock64();                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += prof_t1 - prof_t0; \
        prof_t0 = prof_t1;                                                    \
    } while (0)
#else
#define PROF_T0
#define PROF_ADD(slot)
#endif
#ifndef KOPT_U
#define KOPT_U 1  // rows (16-B loads) requested ahead in the bilinear forms; more only spills registers (measured 8: 8.3, 4: 7.2, 2: 6.9, 1: 6.6 ms)
#endif
#define KOPT_STRIDE 12  // doubles per pair in the optimiser's result row (see k_get_h)
#define PI 3.141592653589793238462643383279502884
#define PISQ (PI * PI)

// ---- small fp64 GEMM: C[i][j] = sum_p A[i][p] B[j][p]  (NT), M=N=K=F, batched ---------------------------------
// EPI 0: plain store.  EPI 1: v = acc / sums[b]; out[j][i] = v*v (transposed, squared).
// 64 x 64 output tile per block, 32 x 32 per wave as 2 x 2 v_mfma_f64_16x16x4_f64 tiles; K is staged through LDS in
// chunks of 32, register-prefetched one step ahead (k-major rows padded to 65 doubles: the 16 lanes of a k-group read consecutive doubles).  fp64 MFMA
// runs at the fp64 vector rate, but one instruction carries 1024 FMAs per wave against two LDS reads per lane, where
// the scalar-FMA tile needed eight -- the old kernel was LDS-bound at a quarter of the fp64 peak.
// Operand / result lanes (CDNA4): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 r][col = lane & 15] for result register r.
typedef double gd_f64x4 __attribute__((ext_vector_type(4)));
template <int EPI>
__global__ void __launch_bounds__(256) k_gemm_nt(const double* __restrict__ A, int64_t strideA,
                                                 const double* __restrict__ Bm, int64_t strideB, int F,
                                                 double* __restrict__ Cm, int64_t strideC,
                                                 const double* __restrict__ sums) {
    constexpr int KC = 32;  // K rows staged per step
    __shared__ double As[KC][65];
    __shared__ double Bs[KC][65];
    const int b = blockIdx.z;
    const double* Ab = A + (int64_t)b * strideA;
    const double* Bb = Bm + (int64_t)b * strideB;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int l15 = lane & 15, lk = lane >> 4;
    gd_f64x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = (gd_f64x4){0.0, 0.0, 0.0, 0.0};
    // each thread stages 8 elements of each operand per step: element q -> (row ii = e >> 5, k = e & 31), e = tid + 256 q;
    // the next step's global loads are issued before this step's MFMAs so that they overlap
    constexpr int NQ = 64 * KC / 256;
    double ra[NQ], rb[NQ];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = threadIdx.x + 256 * q;
            const int pp = e & (KC - 1), ii = e / KC;
            const int p = p0 + pp;
            ra[q] = (i0 + ii < F && p < F) ? Ab[(int64_t)(i0 + ii) * F + p] : 0.0;
            rb[q] = (j0 + ii < F && p < F) ? Bb[(int64_t)(j0 + ii) * F + p] : 0.0;
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < F; p0 += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = threadIdx.x + 256 * q;
            As[e & (KC - 1)][e / KC] = ra[q];
            Bs[e & (KC - 1)][e / KC] = rb[q];
        }
        __syncthreads();
        if (p0 + KC < F) fetch(p0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            const int k = kk * 4 + lk;
            const double a0 = As[k][wi + l15], a1 = As[k][wi + 16 + l15];
            const double b0 = Bs[k][wj + l15], b1 = Bs[k][wj + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    double* Cb = Cm + (int64_t)b * strideC;
    const double inv = (EPI == 1) ? sums[b] : 1.0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi + u * 16 + lk + 4 * r, j = j0 + wj + v * 16 + l15;
                if (i < F && j < F) {
                    if (EPI == 0) {
                        Cb[(int64_t)i * F + j] = acc[u][v][r];
                    } else {
                        const double val = acc[u][v][r] / inv;
                        Cb[(int64_t)j * F + i] = val * val;
                    }
                }
            }
}


// ---- DCT-II of the rows of a batch of F x F matrices through a half-length complex transform in LDS ---------------------
// y[k] = 2 sum_n x[n] cos(pi k (2n+1) / (2F))  (scipy.fftpack.dct type 2, unnormalised) by Makhoul's reordering:
// v[m] = x[2m] (m < F/2), v[F-1-m] = x[2m+1]; V = DFT_F(v) from ONE complex transform of length H = F/2 of the packed
// sequence z[n] = v[2n] + i v[2n+1] (the real-input un-mixing of k_rows_fwd); y[k] = 2 Re(c_k V[k]), c_k = e^{-i pi k / 2F},
// and y[F-k] = 2 Re(c_{F-k} conj V[k]).  The result is stored TRANSPOSED: dst[b][k][row], so that two passes make the 2D
// transform (pass 1: along x, pass 2: along y) -- 2 x (read + write) of the matrix instead of two F^3 GEMMs (which ran at
// 37 TFLOP/s of fp64 MFMA: 2.0 ms per 1225 pairs; this: memory-bound).
// EPI 1 (second pass): v = y / sums[b]; out = v * v  (the squared, normalised coefficients the fixed point works on).
// grid (ceil(F / RPB), B), RPB * 32 threads: RPB rows per block, one 32-lane group per row (ldsfft.hpp).
// dynamic LDS: F twiddles e^{-2 pi i k / F} | F post-twiddles c_k | RPB x (H complex transform buffer + F doubles staging)
template <int EPI, bool BIG>
__global__ void __launch_bounds__(512) k_dct_pass(const double* __restrict__ src, int F, FftDev plH, const double2* __restrict__ twg,
                                                  const double2* __restrict__ cg, const double* __restrict__ sums,
                                                  double* __restrict__ dst) {
    extern __shared__ double2 dsh[];
    const int H = F >> 1, b = blockIdx.y, RPB = blockDim.x / FT;
    double2* tw = dsh;
    double2* ck = dsh + F;
    const int g = threadIdx.x / FT, t = threadIdx.x % FT;
    double2* buf = dsh + 2 * F + (size_t)g * F;  // H complex
    double* stage = (double*)(buf + H);           // F doubles
    const int row = blockIdx.x * RPB + g;
    const bool active = row < F;
    {
        // a lane's loads -- eight elements of its row, its share of the two tables -- are all requested before the first is
        // stored (clamped addresses, unconditional stores: the plain loops waited for every load before the next was issued,
        // eight memory latencies in sequence per block; ISA reading, round 5)
        const double* x = src + (int64_t)b * F * F + (int64_t)min(row, F - 1) * F;
        for (int i0 = t; i0 < F; i0 += 8 * FT) {
            double xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = x[min(i0 + q * FT, F - 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) stage[min(i0 + q * FT, F - 1)] = xv[q];
        }
        for (int i0 = threadIdx.x; i0 < F; i0 += 2 * blockDim.x) {
            const int i1 = min(i0 + (int)blockDim.x, F - 1);
            const double2 a0 = twg[i0], a1 = twg[i1], c0 = cg[i0], c1 = cg[i1];
            __builtin_amdgcn_sched_barrier(0);
            tw[i0] = a0, tw[i1] = a1, ck[i0] = c0, ck[i1] = c1;
        }
    }
    __syncthreads();
    if (active) {
        auto v = [&](int m) { return m < H ? stage[2 * m] : stage[2 * (F - 1 - m) + 1]; };
        for (int n = t; n < H; n += FT) buf[n] = make_double2(v(2 * n), v(2 * n + 1));
    }
    group_sync();
    fft_full<BIG, false>(buf, tw, 2, plH, t);
    if (active) {
        const double inv = EPI == 1 ? sums[b] : 1.0;
        auto put = [&](int k, double y) {
            if (EPI == 1) {
                const double val = y / inv;
                stage[k] = val * val;
            } else {
                stage[k] = y;
            }
        };
        for (int k = t; 2 * k <= H; k += FT) {
            if (k == 0) {
                const double2 z = buf[0];
                put(0, 2.0 * (z.x + z.y));
                const double vh = z.x - z.y;  // V[H], real
                put(H, 2.0 * (ck[H].x * vh));
                continue;
            }
            const double2 zk = buf[k], zm = buf[H - k];
            const double ax = zk.x + zm.x, ay = zk.y - zm.y, bx = zk.x - zm.x, by = zk.y + zm.y;
            {
                const double2 e = tw[k];
                const double u = fma(e.x, bx, -(e.y * by)), w = fma(e.x, by, e.y * bx);
                const double vx = 0.5 * (ax + w), vy = 0.5 * (ay - u);  // V[k]
                const double2 c1 = ck[k], c2 = ck[F - k];
                put(k, 2.0 * fma(c1.x, vx, -(c1.y * vy)));
                put(F - k, 2.0 * fma(c2.x, vx, c2.y * vy));
            }
            if (2 * k < H) {  // V[H - k] from the same two values
                const double2 f = tw[H - k];
                const double u2 = -fma(f.x, bx, f.y * by), w2 = fma(f.x, by, -(f.y * bx));
                const double vx = 0.5 * (ax + w2), vy = 0.5 * (-ay - u2);
                const double2 c1 = ck[H - k], c2 = ck[H + k];
                put(H - k, 2.0 * fma(c1.x, vx, -(c1.y * vy)));
                put(H + k, 2.0 * fma(c2.x, vx, c2.y * vy));
            }
        }
    }
    __syncthreads();
    // transposed store: the block's RPB rows are RPB consecutive doubles of every output row k
    double* out = dst + (int64_t)b * F * F + (int64_t)blockIdx.x * RPB;
    const int nrows = min(RPB, F - blockIdx.x * RPB);
    for (int e = threadIdx.x; e < F * RPB; e += blockDim.x) {
        const int k = e / RPB, r = e - k * RPB;
        if (r < nrows) {
            const double* st = (const double*)(dsh + 2 * F + (size_t)r * F + H);
            out[(int64_t)k * F + r] = st[k];
        }
    }
}

// DCT-II matrix: D[k][n] = 2 cos(pi k (2n+1) / (2F))   (scipy.fftpack.dct type 2, unnormalised)
__global__ void k_dct_matrix(int F, double* __restrict__ D) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= F * F) return;
    const int k = e / F, n = e % F;
    const long long m = ((long long)k * (2 * n + 1)) % (4LL * F);  // angle = pi*m/(2F)
    const int q = (int)(m / F);
    const int r = (int)(m - (long long)q * F);
    const double c = cospi((double)r / (2.0 * F)), s = sinpi((double)r / (2.0 * F));
    double v = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    D[e] = 2.0 * v;
}

// per-pair histogram sums
__global__ void k_hist_sums(const double* __restrict__ hist, int FF, double* __restrict__ sums) {
    __shared__ double red[16];
    const double* h = hist + (int64_t)blockIdx.x * FF;
    double s = 0;
    for (int i = threadIdx.x; i < FF; i += blockDim.x) s += h[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sums[blockIdx.x] = s;
}

// gather + normalise the histograms that need the power spectrum
__global__ void k_gather_norm(const double* __restrict__ hist, const int* __restrict__ which, const double* __restrict__ sums,
                              int FF, double* __restrict__ out) {
    const int b = which[blockIdx.y];
    const double s = sums[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < FF; i += gridDim.x * blockDim.x)
        out[(int64_t)blockIdx.y * FF + i] = hist[(int64_t)b * FF + i] / s;
}

// full F x F power spectrum |Z|^2 from the hermitian half Z[F][F/2+1]
__global__ void k_power_full(const double2* __restrict__ Z, int F, double* __restrict__ out) {
    const int Fh = F / 2 + 1;
    const double2* Zb = Z + (int64_t)blockIdx.y * F * Fh;
    double* ob = out + (int64_t)blockIdx.y * F * F;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * F; e += gridDim.x * blockDim.x) {
        const int i = e / F, j = e % F;
        double2 z;
        if (j < Fh)
            z = Zb[(int64_t)i * Fh + j];
        else
            z = Zb[(int64_t)((F - i) % F) * Fh + (F - j)];
        ob[e] = z.x * z.x + z.y * z.y;
    }
}

// ---- the per-pair solver ----------------------------------------------------------------------------------------
struct KoptLds {
    double* wx;     // MAXF x F
    double* wy;     // MAXF x F
    double* res;    // MAXF results + scratch
    double* red;    // 16
    double* times;  // MAXF plug-in times of the current level (one thread computes each; pow() is expensive)
    double* scale;  // per-level output factor: (-1)^L pi^(2L) / 4 for the even, (2 pi)^L for the odd functionals
    double* tile;   // staging area for rows of the matrix (LDS-DMA target), nullptr when F is not a multiple of 128
    int tile_doubles;
};

// Evaluate m bilinear forms  sum_ij wy_q[i] M[i][j] wx_q[j]  for the weight sets currently in LDS.
// Only rows / columns below kmax are visited: beyond it every weight is < 1e-30 of its form's maximum (see
// weight_cutoff), which cannot change the fp64 result.
__device__ void bilinear_forms_reg(const double* __restrict__ M, int F, int m, KoptLds L, int kmax);

// The matrix (512 KB at F = 256) is re-read ~25 times per pair and 256 pairs are in flight, so it comes from MALL / HBM
// at ~2 us per access: with register loads a block has 16 KB in flight and the launch crawls at 1.8 TB/s (PMC: 89 % L2
// misses, 15.5 GB per 1200 pairs).  Here whole rows go straight into LDS (global_load_lds_dwordx4: one 1-KB row segment
// per wave instruction, no VGPRs), up to 128 KB per tile in flight; the forms are then accumulated from LDS.
// (Tried and measured slower: two 64-KB buffers with asm-issued loads and hand-placed s_waitcnt vmcnt(4) so that tile
// t+1 streams in while tile t is accumulated -- 11.4 vs 5.8 ms for 1200 pairs: twice the barriers per pass cost more
// than the overlap gains; the fallbacks as noinline calls -- 8.4 ms; and barrier-free wave-private row rings (each wave
// streams its own rows with counted vmcnt waits) -- 7.6 ms, while merely carrying that extra code path slowed THIS one
// from 5.8 to 7.8 ms: the kernel sits at the 128-VGPR limit with ~650 B of scratch per lane, and every variant that adds
// live state or code pays for it in spills.  The next step is to split the solver into smaller kernels, not to tune
// this loop further.)
__device__ void bilinear_forms(const double* __restrict__ M, int F, int m, KoptLds L, int kmax) {
    if (L.tile == nullptr) {
        bilinear_forms_reg(M, F, m, L, kmax);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = (kmax + 127) >> 7;  // 128-column (1 KB) segments of a row that carry weight
    const int rows_fit = L.tile_doubles / (nseg * 128);
    const int jx = threadIdx.x & 127, g = threadIdx.x >> 7;  // 128 column pairs x 8 row groups
    double val[MAXF];
#pragma unroll
    for (int q = 0; q < MAXF; ++q) val[q] = 0;
    PROF_T0;
    for (int r0 = 0; r0 < kmax; r0 += rows_fit) {
        const int nr = min(rows_fit, kmax - r0);
        for (int idx = wave; idx < nr * nseg; idx += KT / 64) {
            const int row = idx / nseg, seg = idx - row * nseg;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(M + (int64_t)(r0 + row) * F + seg * 128 + lane * 2),
                (__attribute__((address_space(3))) void*)(L.tile + idx * 128), 16, 0, 0);
        }
        __syncthreads();  // carries the vmcnt(0) that lands the tile
        PROF_ADD(1);
        for (int j0 = 0; j0 < nseg * 128; j0 += 256) {
            const int j = j0 + 2 * jx;
            if (j < kmax) {
                double ax[MAXF], ay[MAXF];
#pragma unroll
                for (int q = 0; q < MAXF; ++q) ax[q] = ay[q] = 0;
                const double* col = L.tile + (j >> 7) * 128 + (j & 127);
                // two rows per step: the row weights of a form are adjacent, so one 16-B broadcast read serves both
                // (with one row per step the phase was bound by the LDS issue rate, not by the FMAs)
                for (int r = 2 * g; r < nr; r += 2 * (KT / 128)) {
                    const double2 a0 = *reinterpret_cast<const double2*>(col + r * nseg * 128);
                    const bool two = r + 1 < nr;
                    const double2 a1 = two ? *reinterpret_cast<const double2*>(col + (r + 1) * nseg * 128)
                                           : make_double2(0.0, 0.0);
#pragma unroll
                    for (int q = 0; q < MAXF; ++q)
                        if (q < m) {
                            // r0 + r is even and F is even: 16-B aligned
                            const double2 wyq = *reinterpret_cast<const double2*>(L.wy + q * F + r0 + r);
                            ax[q] = fma(wyq.x, a0.x, ax[q]);
                            ay[q] = fma(wyq.x, a0.y, ay[q]);
                            const double wy1 = two ? wyq.y : 0.0;  // never 0 x NaN from the neighbouring words
                            ax[q] = fma(wy1, a1.x, ax[q]);
                            ay[q] = fma(wy1, a1.y, ay[q]);
                        }
                }
#pragma unroll
                for (int q = 0; q < MAXF; ++q)
                    if (q < m) {
                        val[q] = fma(ax[q], L.wx[q * F + j], val[q]);
                        if (j + 1 < kmax) val[q] = fma(ay[q], L.wx[q * F + j + 1], val[q]);
                    }
            }
        }
        __syncthreads();  // the tile is consumed before the next one overwrites it
        PROF_ADD(2);
    }
    const int wv = wave;
#pragma unroll
    for (int q = 0; q < MAXF; ++q)
        if (q < m) {
            const double r = wave_sum(val[q]);
            if (lane == 0) L.wx[q * 16 + wv] = r;
        }
    __syncthreads();
    if ((int)threadIdx.x < m) {
        double r = 0;
        for (int i = 0; i < KT / 64; ++i) r += L.wx[threadIdx.x * 16 + i];
        L.res[threadIdx.x] = r;
    }
    __syncthreads();
    PROF_ADD(3);
}

__device__ void bilinear_forms_reg(const double* __restrict__ M, int F, int m, KoptLds L, int kmax) {
    // 8 row groups x 128 column pairs; the matrix comes from L2 / MALL with ~1 us latency, so what matters is the number
    // of independent loads in flight: 8 rows (16-B loads) are requested before the first is consumed.
    constexpr int U = KOPT_U;
    const int jx = threadIdx.x & 127, g = threadIdx.x >> 7;
    const int rows_per = (kmax + KT / 128 - 1) / (KT / 128);
    const int r_lo = g * rows_per, r_hi = min(kmax, r_lo + rows_per);
    double val[MAXF];
#pragma unroll
    for (int q = 0; q < MAXF; ++q) val[q] = 0;
    for (int j0 = 0; j0 < kmax; j0 += 256) {
        const int j = j0 + 2 * jx;  // F is even and rows are 16-B aligned
        if (j < kmax) {
            double ax[MAXF], ay[MAXF];
#pragma unroll
            for (int q = 0; q < MAXF; ++q) ax[q] = ay[q] = 0;
            for (int i = r_lo; i < r_hi; i += U) {
                double2 a[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    a[u] = (i + u < r_hi) ? *reinterpret_cast<const double2*>(M + (int64_t)(i + u) * F + j)
                                          : make_double2(0.0, 0.0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int iu = min(i + u, r_hi - 1);  // the padded rows carry a = 0
#pragma unroll
                    for (int q = 0; q < MAXF; ++q)
                        if (q < m) {
                            const double wyq = L.wy[q * F + iu];
                            ax[q] = fma(wyq, a[u].x, ax[q]);
                            ay[q] = fma(wyq, a[u].y, ay[q]);
                        }
                }
            }
#pragma unroll
            for (int q = 0; q < MAXF; ++q)
                if (q < m) {
                    val[q] = fma(ax[q], L.wx[q * F + j], val[q]);
                    if (j + 1 < kmax) val[q] = fma(ay[q], L.wx[q * F + j + 1], val[q]);
                }
        }
    }
    // all m block sums with two barriers: per-wave partials -> LDS (wx is free now) -> one thread per form adds them
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < MAXF; ++q)
        if (q < m) {
            const double r = wave_sum(val[q]);
            if (lane == 0) L.wx[q * 16 + wv] = r;
        }
    __syncthreads();
    if ((int)threadIdx.x < m) {
        double r = 0;
        for (int i = 0; i < KT / 64; ++i) r += L.wx[threadIdx.x * 16 + i];
        L.res[threadIdx.x] = r;
    }
    __syncthreads();
}

// An index beyond which w(k) = k^(2s) exp(-k^2 pi^2 t) is below e^-69 (1e-30) of its maximum for every derivative
// order s <= Lsum and every plug-in time of the level.  Conservative closed form: the maximum is at least
// w(1) = exp(-c), c = pi^2 t, and ln k^2 <= ln F^2, so  k^2 >= (69 + c + s ln F^2) / c  suffices.
__device__ __forceinline__ int weight_cutoff(int F, int Lsum, int m, const double* times) {
    double tmin = times[0];
    for (int q = 1; q < m; ++q) tmin = fmin(tmin, times[q]);
    const double c = PISQ * tmin;
    if (!(c > 0.0)) return F;  // t = 0 (Brent's left end), negative or NaN plug-in times: no truncation
    const double k2 = (69.0 + (double)Lsum * 2.0 * log((double)F)) / c + 1.0;
    if (!(k2 < (double)F * (double)F)) return F;
    const int k = (int)sqrt(k2) + 2;
    return k < F ? k : F;
}

// even functionals: psi([s0,s1], time) for the forms of one level (all with s0+s1 = Lsum); times in L.times
__device__ void psi_level(const double* __restrict__ SQ, int F, int Lsum, double* out, KoptLds L) {
    const int m = Lsum + 1;  // forms [a, Lsum-a], a = 0..Lsum   (m <= MAXF)
    PROF_T0;
    __syncthreads();
    PROF_ADD(4);
    for (int e = threadIdx.x; e < m * F; e += KT) {
        const int q = e / F, k = e % F;
        double vx = 0, vy = 0;
        if (k > 0) {
            const double I = (double)k * (double)k;
            const double logI = log(I);
            const double w = -I * (PISQ * L.times[q]);
            vx = exp(w + logI * (double)q);
            vy = exp(w + logI * (double)(Lsum - q));
        }
        L.wx[e] = vx;
        L.wy[e] = vy;
    }
    const int kmax = weight_cutoff(F, Lsum, m, L.times);
    __syncthreads();
    PROF_ADD(0);
#ifdef KOPT_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[5] += 1, g_prof[6] += kmax;
#endif
    bilinear_forms(SQ, F, m, L, kmax);
    const double sc = L.scale[Lsum];
    for (int q = 0; q < m; ++q) out[q] = (Lsum & 1 ? -1.0 : 1.0) * L.res[q] * sc / 4.0;
    __syncthreads();
}

// kde_bandwidth.py:140-143
__device__ __forceinline__ double odd_dfact(int j) {  // prod(arange(1, 2j, 2)) = (2j-1)!!
    double p = 1.0;
    for (int q = 1; q < 2 * j; q += 2) p *= (double)q;
    return p;
}
__device__ __forceinline__ double k_even(int j) {
    const double v = odd_dfact(j) / sqrt(2.0 * PI);
    return (j == 0) ? 1.0 / sqrt(2.0 * PI) : ((j & 1) ? -v : v);
}
__device__ __forceinline__ double k_odd(int j) {
    return (j == 0) ? 1.0 : odd_dfact(j) / pow(2.0, (double)(j + 1)) / sqrt(PI);
}

// func2d for every [a, L-a], L = 5 .. Lmin (memoised recursion of kde_bandwidth.py:188-196).
// lev[L][a] receives func2d([a, L-a], t).
__device__ void func2d_levels(const double* __restrict__ SQ, int F, double N, double t, int Lmin, double lev[6][MAXF],
                              KoptLds L) {
    __syncthreads();
    if (threadIdx.x < MAXF) L.times[threadIdx.x] = t;
    psi_level(SQ, F, 5, lev[5], L);
    for (int Ls = 4; Ls >= Lmin; --Ls) {
        PROF_T0;
        if ((int)threadIdx.x <= Ls) {  // one thread per form evaluates its plug-in time (kde_bandwidth.py:191-193)
            const int a = threadIdx.x;
            const double cst = (1.0 + pow(0.5, (double)(Ls + 1))) / 3.0;
            const double sum_func = lev[Ls + 1][a + 1] + lev[Ls + 1][a];
            L.times[a] = pow(-2.0 * cst * k_even(a) * k_even(Ls - a) / N / sum_func, 1.0 / (2.0 + Ls));
        }
        PROF_ADD(7);
        psi_level(SQ, F, Ls, lev[Ls], L);
    }
}

__device__ double fixed_point_2d(const double* __restrict__ SQ, int F, double N, double t, double lev[6][MAXF],
                                 KoptLds L) {
    func2d_levels(SQ, F, N, t, 2, lev, L);
    const double sum_func = lev[2][0] + lev[2][2] + 2.0 * lev[2][1];
    const double time = pow(2.0 * PI * N * sum_func, -1.0 / 3.0);
    return (t - time) / time;
}

// odd functionals: psi_odd for forms [1+2q, Lsum-1-2q]; times in L.times
__device__ void psi_odd_level(const double* __restrict__ PW, int F, int Lsum, int m, double* out, KoptLds L) {
    __syncthreads();
    for (int e = threadIdx.x; e < m * F; e += KT) {
        const int q = e / F, k = e % F;
        const double f = (k <= (F - 1) / 2) ? (double)k : (double)(k - F);
        const double w = exp(-(f * f) * (4.0 * PISQ * L.times[q]));
        const int s0 = 1 + 2 * q, s1 = Lsum - s0;
        L.wx[e] = w * pow(f, (double)s0);
        L.wy[e] = w * pow(f, (double)s1);
    }
    __syncthreads();
    bilinear_forms(PW, F, m, L, F);
    const double sc = L.scale[6 + Lsum / 2];
    for (int q = 0; q < m; ++q) out[q] = L.res[q] * sc;
    __syncthreads();
}

struct KoptPair {
    double N, fallback_t;
    int do_corr, pw_index;
};

__global__ void __launch_bounds__(KT) k_kopt2d(const double* __restrict__ SQ_all, const double* __restrict__ PW_all,
                                               const KoptPair* __restrict__ pairs, int F, int tile_doubles,
                                               double* __restrict__ out) {
    extern __shared__ double lds[];
    KoptLds L;
    L.wx = lds;
    L.wy = lds + MAXF * F;
    L.res = L.wy + MAXF * F;
    L.red = L.res + 8;
    L.times = L.red + 16;
    L.scale = L.times + 8;
    L.tile = tile_doubles > 0 ? L.scale + 16 : nullptr;
    L.tile_doubles = tile_doubles;
    if (threadIdx.x < 6) L.scale[threadIdx.x] = pow(PI, (double)(2 * threadIdx.x));               // even: pi^(2L)
    if (threadIdx.x >= 8 && threadIdx.x < 14) L.scale[threadIdx.x - 2] = pow(2.0 * PI, (double)(2 * (threadIdx.x - 8)));  // odd: (2pi)^L, L = 0,2,..,10
    __syncthreads();
    const int b = blockIdx.x;
    const KoptPair P = pairs[b];
    const double* SQ = SQ_all + (int64_t)b * F * F;
    const double N = P.N;
    double lev[6][MAXF];
    for (int a = 0; a < 6; ++a)
        for (int q = 0; q < MAXF; ++q) lev[a][q] = 0;

    // ---- Brent's method (scipy.optimize.brentq: xtol=1e-6, rtol=4eps, maxiter=100) on [0, 0.1] ----
    const double xtol = 0.001 * 0.001, rtol = 4.0 * 2.220446049250313e-16;
    double xpre = 0.0, xcur = 0.1, xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
    double fpre = fixed_point_2d(SQ, F, N, xpre, lev, L);
    double fcur = fixed_point_2d(SQ, F, N, xcur, lev, L);
    double t_star = 0.0;
    int status = GD_OK;
    bool done = false;
    if (fpre != fpre || fcur != fcur) {
        status = GD_ERR_SOLVER;
        done = true;
    } else if (fpre == 0) {
        t_star = xpre;
        done = true;
    } else if (fcur == 0) {
        t_star = xcur;
        done = true;
    } else if (signbit(fpre) == signbit(fcur)) {
        status = GD_ERR_SOLVER;  // "f(a) and f(b) must have different signs"
        done = true;
    }
    if (!done) {
        status = GD_ERR_SOLVER;  // convergence error unless we return inside the loop
        for (int it = 0; it < 100; ++it) {
            if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
                xblk = xpre;
                fblk = fpre;
                spre = scur = xcur - xpre;
            }
            if (fabs(fblk) < fabs(fcur)) {
                xpre = xcur, xcur = xblk, xblk = xpre;
                fpre = fcur, fcur = fblk, fblk = fpre;
            }
            const double delta = (xtol + rtol * fabs(xcur)) / 2;
            const double sbis = (xblk - xcur) / 2;
            if (fcur == 0 || fabs(sbis) < delta) {
                t_star = xcur;
                status = GD_OK;
                break;
            }
            if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
                double stry;
                if (xpre == xblk) {
                    stry = -fcur * (xcur - xpre) / (fcur - fpre);
                } else {
                    const double dpre = (fpre - fcur) / (xpre - xcur);
                    const double dblk = (fblk - fcur) / (xblk - xcur);
                    stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
                }
                if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) {
                    spre = scur;
                    scur = stry;
                } else {
                    spre = sbis;
                    scur = sbis;
                }
            } else {
                spre = sbis;
                scur = sbis;
            }
            xpre = xcur;
            fpre = fcur;
            if (fabs(scur) > delta)
                xcur += scur;
            else
                xcur += (sbis > 0 ? delta : -delta);
            fcur = fixed_point_2d(SQ, F, N, xcur, lev, L);
            if (fcur != fcur) break;  // NaN: scipy would wander to a convergence error
        }
    }
    const bool have_fb = P.fallback_t > 0;
    if (status == GD_OK) {
        if (have_fb && t_star > 0.01 && t_star > 2 * P.fallback_t) t_star = P.fallback_t;  // kde_bandwidth.py:164-167
    } else if (have_fb) {
        t_star = P.fallback_t;  // kde_bandwidth.py:168-173
        status = GD_OK;
    }
    double p02 = NAN, p20 = NAN, p11 = NAN, p00 = NAN, p13 = NAN, p31 = NAN;
    if (status == GD_OK) {
        func2d_levels(SQ, F, N, t_star, P.do_corr ? 0 : 2, lev, L);
        p02 = lev[2][0], p20 = lev[2][2], p11 = lev[2][1];
        if (P.do_corr) {
            p00 = lev[0][0];
            const double* PW = PW_all + (int64_t)P.pw_index * F * F;
            double odd[4][MAXF];
            __syncthreads();
            if (threadIdx.x < MAXF) L.times[threadIdx.x] = t_star;
            psi_odd_level(PW, F, 10, 5, odd[3], L);  // [1,9],[3,7],[5,5],[7,3],[9,1]
            int li = 2;
            for (int Ls = 8; Ls >= 4; Ls -= 2, --li) {
                const int m = Ls / 2;  // forms [1+2q, Ls-1-2q]
                if ((int)threadIdx.x < m) {
                    const int q = threadIdx.x;
                    const double cst = 8.0 * (1.0 - pow(2.0, (double)(-Ls - 1))) / 3.0;
                    const int s0 = 1 + 2 * q, s1 = Ls - s0;
                    // func2d_odd([s0+2,s1]) + func2d_odd([s0,s1+2]) : entries q+1 and q of the level above
                    const double sum_func = odd[li + 1][q + 1] + odd[li + 1][q];
                    L.times[q] = pow(cst * p00 * k_odd(s0) * k_odd(s1) / (N * N) / (sum_func * sum_func), 1.0 / (3.0 + Ls));
                }
                psi_odd_level(PW, F, Ls, m, odd[li], L);
            }
            p13 = odd[0][0];  // [1,3]
            p31 = odd[0][1];  // [3,1]
        }
    }
    if (threadIdx.x == 0) {
        double* o = out + (int64_t)b * KOPT_STRIDE;
        o[0] = (status == GD_OK) ? t_star : NAN;
        o[1] = p02, o[2] = p20, o[3] = p11, o[4] = p00, o[5] = p13, o[6] = p31, o[7] = (double)status;
    }
}

#ifdef KOPT_PROFILE
#define RPROF_T0 long long prof_t0 = clock64()
#define RPROF_ADD(slot)                                                              \
    do {                                                                             \
        const long long prof_t1 = clock64();                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) V.sh->prof[slot] += prof_t1 - prof_t0; \
        prof_t0 = prof_t1;                                                           \
    } while (0)
#else
#define RPROF_T0
#define RPROF_ADD(slot)
#endif
// ---- the same solver with the matrix RESIDENT on the CU (F = 256) ----------------------------------------------------
// k_kopt2d streams the 512-KB matrix of a pair through LDS ~25 times (6-7 Brent evaluations x 4 dependent levels).  Here
// the block loads it ONCE: 512 threads, wave w owns rows [32w, 32w+32), lane l the four columns {2l, 2l+1, 128+2l,
// 129+2l} of each -- 24 of those rows live in the lane's registers (96 doubles = 192 VGPRs of the 256 a wave has at two
// waves per SIMD), the other 8 in LDS (8 waves x 8 rows x 2 KB = 128 KB) next to the weight tables (24 KB).  A level is
// then m x (128 FMAs from registers/LDS + 16 broadcast reads of the row weights) per lane and one block reduction; the
// scalar control flow (scipy's brentq, the plug-in times) is run by thread 0 / thread a on state kept in LDS so that no
// lane carries it across the passes.  Same formulas and evaluation points as k_kopt2d; only the order in which the
// 65 536 products of a form are added differs (per column over the lane's 32 rows, then columns, lanes, waves).
#define KR_F 256
#define KR_T 512
#define KR_REG 24  // rows of a wave's 32 held in registers
#define KR_LDS 8   // ... and in LDS

struct KresShared {
    double part[8 * MAXF];
    double times[8];
    double scale[16];
    double lev[6][MAXF];
    double odd[4][MAXF];
    double N, fallback_t;
    double x, f;  // the evaluation point handed to the block, the function value handed back
    double xpre, xcur, xblk, fpre, fcur, fblk, spre, scur, t_star;
    double t;             // the time the current chain of levels is evaluated at
    double p00;           // psi_00 (odd launch)
    double ceven[5][MAXF];  // -2 cst(Ls) K(a) K(Ls-a)
    double kodd[10], codd[3];
    int status, done, it, do_corr;
    int stage, Ls, Lmin;
#ifdef KOPT_PROFILE
    long long prof[8], wave_cycles[8];
#endif
};

struct KresView {
    double* wxs;   // MAXF x 256
    double* wys;   // MAXF x 256
    double* mt;    // 8 waves x KR_LDS rows x 256
    double* logI;  // 256: log(k^2)
    KresShared* sh;
};

__device__ __forceinline__ void kres_load(const double* __restrict__ M, double (&m)[KR_REG][4], double* mt, int w, int l) {
    const double* base = M + (int64_t)(32 * w) * KR_F + 2 * l;
#pragma unroll
    for (int r = 0; r < KR_REG; ++r) {
        const double2 a = *reinterpret_cast<const double2*>(base + r * KR_F);
        const double2 b = *reinterpret_cast<const double2*>(base + r * KR_F + 128);
        m[r][0] = a.x, m[r][1] = a.y, m[r][2] = b.x, m[r][3] = b.y;
    }
    double* row = mt + (w * KR_LDS) * KR_F + 2 * l;
#pragma unroll
    for (int r = 0; r < KR_LDS; ++r) {
        const double2 a = *reinterpret_cast<const double2*>(base + (KR_REG + r) * KR_F);
        const double2 b = *reinterpret_cast<const double2*>(base + (KR_REG + r) * KR_F + 128);
        *reinterpret_cast<double2*>(row + r * KR_F) = a;
        *reinterpret_cast<double2*>(row + r * KR_F + 128) = b;
    }
}

// acc += y[lane J of the 16-lane row] * m : the row weight is broadcast inside the instruction (DPP row_newbcast), so a
// form's 32 row weights cost two LDS reads per lane instead of sixteen wave-uniform ones in front of every eight FMAs
// (measured: the FMA phase of a level fell from 10.7 to 5.5 k cycles; the compiler does not form DPP for f64 by itself).
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, double y, double m) {
#ifdef KOPT_PROFILE  // pinned between the cycle-counter reads
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(m), "n"(J));
#else
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(m), "n"(J));
#endif
}
#define KR_ROW(R, C0, C1, C2, C3, M0, M1, M2, M3)        \
    fmac_bc<(R) & 15>(C0, (R) < 16 ? y0 : y1, M0);      \
    fmac_bc<(R) & 15>(C1, (R) < 16 ? y0 : y1, M1);      \
    fmac_bc<(R) & 15>(C2, (R) < 16 ? y0 : y1, M2);      \
    fmac_bc<(R) & 15>(C3, (R) < 16 ? y0 : y1, M3);
#define KR_REG_PAIR(R)                                                         \
    KR_ROW(R, c0, c1, c2, c3, m[R][0], m[R][1], m[R][2], m[R][3])              \
    KR_ROW(R + 1, d0, d1, d2, d3, m[R + 1][0], m[R + 1][1], m[R + 1][2], m[R + 1][3])
#define KR_LDS_LOAD(G)                                                                                  \
    const double2 a0_##G = *reinterpret_cast<const double2*>(mt_l + (2 * (G)) * KR_F);                    \
    const double2 b0_##G = *reinterpret_cast<const double2*>(mt_l + (2 * (G)) * KR_F + 128);              \
    const double2 a1_##G = *reinterpret_cast<const double2*>(mt_l + (2 * (G) + 1) * KR_F);                \
    const double2 b1_##G = *reinterpret_cast<const double2*>(mt_l + (2 * (G) + 1) * KR_F + 128);
#define KR_LDS_USE(G)                                                                      \
    KR_ROW(KR_REG + 2 * (G), c0, c1, c2, c3, a0_##G.x, a0_##G.y, b0_##G.x, b0_##G.y)       \
    KR_ROW(KR_REG + 2 * (G) + 1, d0, d1, d2, d3, a1_##G.x, a1_##G.y, b1_##G.x, b1_##G.y)

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// v + v[lane ^ K] inside each half of the wave: ds_swizzle carries its pattern in the instruction (no address register,
// no LDS memory touched), so the five steps of a form's lane sum ride on the LDS pipe between the FMAs of the NEXT form.
template <int K>
__device__ __forceinline__ double add_xor(double v) {
    constexpr int pat = 0x1f | (K << 10);  // bit mode: and 0x1f, or 0, xor K
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return v + __hiloint2double(hi, lo);
}
// the same step cut in two so that its LDS round trip lies under a few rows of FMAs; sched_barrier pins both halves where
// they are written (left alone, the scheduler starts the ladder at the top of the loop and waits for every step)
#define KR_SW_ISSUE(K)                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    slo = __builtin_amdgcn_ds_swizzle(__double2loint(vp), 0x1f | ((K) << 10));    \
    shi = __builtin_amdgcn_ds_swizzle(__double2hiint(vp), 0x1f | ((K) << 10));    \
    __builtin_amdgcn_sched_barrier(0);
#define KR_SW_ADD()                            \
    __builtin_amdgcn_sched_barrier(0);         \
    vp = vp + __hiloint2double(shi, slo);      \
    __builtin_amdgcn_sched_barrier(0);

// mforms bilinear forms with the weights in wxs / wys: per-wave partial sums into sh->part (the caller adds the eight).
// (Measured on the way here, cycles of block 0 per level: wave-uniform LDS reads of the row weights in front of every
// eight FMAs 10.7 k; DPP broadcast 5.5 k but a DPP/readlane lane sum per form added 3.3 k of waiting at the barrier.)
__device__ __forceinline__ void kres_forms(const double (&m)[KR_REG][4], KresView V, int mforms, int w, int l) {
    static_assert(KR_REG == 24 && KR_LDS == 8, "the row lists below are written out for 24 + 8 rows");
    const double* mt_l = V.mt + (w * KR_LDS) * KR_F + 2 * l;
    RPROF_T0;
#ifdef KOPT_PROFILE
    const long long wave_t0 = clock64();
#endif
    const double* wy = V.wys + 32 * w + (l & 15);
    double y0n = wy[0], y1n = wy[16];  // lane 16k + j holds the weights of rows j and 16 + j
    double vp = 0;                      // the previous form's per-lane value, summed over the lanes during this form
#pragma unroll 1
    for (int q = 0; q <= mforms; ++q) {
        if (q == mforms) {  // drain: the last form's lane sum
            vp = add_xor<1>(vp), vp = add_xor<2>(vp), vp = add_xor<4>(vp), vp = add_xor<8>(vp), vp = add_xor<16>(vp);
            vp = readlane_f64(vp, 0) + readlane_f64(vp, 32);
            if (l == 0) V.sh->part[w * MAXF + q - 1] = vp;
            break;
        }
        double y0 = y0n, y1 = y1n;
        // a VALU write of a DPP source needs two wait states before the DPP read (the compiler does not see into the asm)
        asm volatile("s_nop 1" : "+v"(y0), "+v"(y1));
        // the rows in LDS are requested a group ahead of their use, between the register rows
        KR_LDS_LOAD(0)
        if (q + 1 < mforms) y0n = wy[(q + 1) * KR_F], y1n = wy[(q + 1) * KR_F + 16];
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        int slo, shi;
        KR_SW_ISSUE(1)
        KR_REG_PAIR(0) KR_REG_PAIR(2)
        KR_SW_ADD() KR_SW_ISSUE(2)
        KR_REG_PAIR(4)
        KR_LDS_USE(0) KR_LDS_LOAD(1)
        KR_REG_PAIR(6)
        KR_SW_ADD() KR_SW_ISSUE(4)
        KR_REG_PAIR(8) KR_REG_PAIR(10)
        KR_LDS_USE(1) KR_LDS_LOAD(2)
        KR_SW_ADD() KR_SW_ISSUE(8)
        KR_REG_PAIR(12) KR_REG_PAIR(14)
        KR_SW_ADD() KR_SW_ISSUE(16)
        KR_REG_PAIR(16)
        KR_LDS_USE(2) KR_LDS_LOAD(3)
        const double2 xa = *reinterpret_cast<const double2*>(V.wxs + q * KR_F + 2 * l);
        const double2 xb = *reinterpret_cast<const double2*>(V.wxs + q * KR_F + 128 + 2 * l);
        KR_REG_PAIR(18)
        KR_SW_ADD()
        KR_REG_PAIR(20) KR_REG_PAIR(22)
        KR_LDS_USE(3)
        vp = readlane_f64(vp, 0) + readlane_f64(vp, 32);
        if (l == 0 && q > 0) V.sh->part[w * MAXF + q - 1] = vp;
        double v = (c0 + d0) * xa.x;
        v = fma(c1 + d1, xa.y, v);
        v = fma(c2 + d2, xb.x, v);
        v = fma(c3 + d3, xb.y, v);
        vp = v;
    }
    RPROF_ADD(2);
#ifdef KOPT_PROFILE
    if (l == 0 && blockIdx.x == 0) V.sh->wave_cycles[w] += clock64() - wave_t0;
#endif
}

// scipy.optimize.brentq's loop (xtol = 1e-6, rtol = 4 eps, maxiter = 100) cut at its function evaluation: consumes the
// value of the last evaluation (sh->f), leaves the next point in sh->x or sets sh->done.  Thread 0 only.
__device__ void kres_brent_step(KresShared* S) {
    const double xtol = 0.001 * 0.001, rtol = 4.0 * 2.220446049250313e-16;
    if (S->it == -2) {  // f(0) arrived
        S->fpre = S->f;
        S->x = S->xcur;
        S->it = -1;
        return;
    }
    if (S->it == -1) {  // f(0.1) arrived
        S->fcur = S->f;
        const double fpre = S->fpre, fcur = S->fcur;
        S->status = GD_OK;
        if (fpre != fpre || fcur != fcur) {
            S->status = GD_ERR_SOLVER, S->done = 1;
        } else if (fpre == 0) {
            S->t_star = S->xpre, S->done = 1;
        } else if (fcur == 0) {
            S->t_star = S->xcur, S->done = 1;
        } else if (signbit(fpre) == signbit(fcur)) {
            S->status = GD_ERR_SOLVER, S->done = 1;  // "f(a) and f(b) must have different signs"
        }
        if (S->done) return;
        S->status = GD_ERR_SOLVER;  // convergence error unless the loop returns
        S->it = 0;
    } else {  // the evaluation that ended iteration it
        S->fcur = S->f;
        if (S->fcur != S->fcur) {  // NaN: scipy would wander to a convergence error
            S->done = 1;
            return;
        }
        S->it += 1;
        if (S->it >= 100) {
            S->done = 1;
            return;
        }
    }
    double xpre = S->xpre, xcur = S->xcur, xblk = S->xblk, fpre = S->fpre, fcur = S->fcur, fblk = S->fblk;
    double spre = S->spre, scur = S->scur;
    if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
        xblk = xpre;
        fblk = fpre;
        spre = scur = xcur - xpre;
    }
    if (fabs(fblk) < fabs(fcur)) {
        xpre = xcur, xcur = xblk, xblk = xpre;
        fpre = fcur, fcur = fblk, fblk = fpre;
    }
    const double delta = (xtol + rtol * fabs(xcur)) / 2;
    const double sbis = (xblk - xcur) / 2;
    if (fcur == 0 || fabs(sbis) < delta) {
        S->t_star = xcur;
        S->status = GD_OK;
        S->done = 1;
        return;
    }
    if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
        double stry;
        if (xpre == xblk) {
            stry = -fcur * (xcur - xpre) / (fcur - fpre);
        } else {
            const double dpre = (fpre - fcur) / (xpre - xcur);
            const double dblk = (fblk - fcur) / (xblk - xcur);
            stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
        }
        if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) {
            spre = scur;
            scur = stry;
        } else {
            spre = sbis;
            scur = sbis;
        }
    } else {
        spre = sbis;
        scur = sbis;
    }
    xpre = xcur;
    fpre = fcur;
    if (fabs(scur) > delta)
        xcur += scur;
    else
        xcur += (sbis > 0 ? delta : -delta);
    S->xpre = xpre, S->xcur = xcur, S->xblk = xblk, S->fpre = fpre, S->fcur = fcur, S->fblk = fblk;
    S->spre = spre, S->scur = scur;
    S->x = xcur;
}

// One code instance of "a level" (plug-in times -> weight tables -> forms -> scaled sums) serves the even chains
// 5 -> Lmin of every fixed-point evaluation, the final chain at t* and the odd chain 10 -> 4 on the power spectrum: the
// kernel is a loop over levels whose next state (level below / next Brent point / final chain / odd chain / exit) is
// decided by thread 0 at the end of a chain.  (With the chains inlined at their seven call sites the compiler carried
// each site's hoisted addresses across the others and spilled ~90 VGPRs, a reload in nearly every basic block.)
enum { KR_STAGE_BRENT = 0, KR_STAGE_FINAL = 1, KR_STAGE_ODD = 2, KR_STAGE_EXIT = 3 };

// The scalar section between two levels is a real call: inlined, its pow() constants were hoisted out of the level loop
// into VGPR pairs and, with 192 VGPRs pinned by the matrix, spilled -- ~18 scratch reloads in front of their uses, each a
// round trip to memory on the serial path of the block (measured: 24 k cycles per level against 8 k).
//
// Called by threads 0..MAXF-1 after the per-wave partial sums of level Ls are in S->part.  Inside a chain, thread a adds
// up the two forms it needs, ([a, Ls-a] and [a+1, Ls-a-1]; each by the same eight additions whoever computes it), and
// leaves the plug-in time of form a of the level below (kde_bandwidth.py:191-193 / 215-219) -- no barrier between "sum"
// and "time".  At the end of a chain thread 0 decides what follows: the next point of Brent's iteration, the final chain
// at t*, or the end.
template <bool ODD>
__device__ __attribute__((noinline)) void kres_tail(KresShared* S, int Ls, int tid) {
    const int mf = ODD ? Ls / 2 : Ls + 1;
    double* row = ODD ? S->odd[Ls / 2 - 2] : S->lev[Ls];
    const double sc = ODD ? S->scale[6 + Ls / 2] : S->scale[Ls];
    auto level_value = [&](int q) {
        double r = 0;
        for (int i = 0; i < KR_T / 64; ++i) r += S->part[i * MAXF + q];
        return ODD ? r * sc : (Ls & 1 ? -1.0 : 1.0) * r * sc / 4.0;
    };
    if (Ls > S->Lmin) {
        const int nLs = Ls - (ODD ? 2 : 1), nmf = ODD ? nLs / 2 : nLs + 1;
        if (tid < nmf) {
            const double la = level_value(tid), la1 = level_value(tid + 1);
            row[tid] = la;
            if (tid == nmf - 1) row[tid + 1] = la1;
            const double sum_func = la1 + la;
            if (!ODD) {
                S->times[tid] = pow(S->ceven[nLs][tid] / S->N / sum_func, 1.0 / (2.0 + nLs));
            } else {
                const int s0 = 1 + 2 * tid, s1 = nLs - s0;
                S->times[tid] = pow(S->codd[nLs / 2 - 2] * S->p00 * S->kodd[s0] * S->kodd[s1] / (S->N * S->N) / (sum_func * sum_func),
                                    1.0 / (3.0 + nLs));
            }
        }
        if (tid == 0) S->Ls = nLs;
        return;
    }
    if (tid != 0) return;
    for (int q = 0; q < mf; ++q) row[q] = level_value(q);
    if (!ODD && S->stage == KR_STAGE_BRENT) {
        const double t = S->t;
        const double sum_func = S->lev[2][0] + S->lev[2][2] + 2.0 * S->lev[2][1];
        const double time = pow(2.0 * PI * S->N * sum_func, -1.0 / 3.0);
        S->f = (t - time) / time;  // fixed_point_2d
        kres_brent_step(S);
        S->Ls = 5;
        if (!S->done) {
            S->t = S->x;
        } else {
            const bool have_fb = S->fallback_t > 0;
            if (S->status == GD_OK) {
                if (have_fb && S->t_star > 0.01 && S->t_star > 2 * S->fallback_t) S->t_star = S->fallback_t;  // kde_bandwidth.py:164-167
            } else if (have_fb) {
                S->t_star = S->fallback_t;  // kde_bandwidth.py:168-173
                S->status = GD_OK;
            }
            S->t = S->t_star;
            S->Lmin = S->do_corr ? 0 : 2;
            S->stage = S->status == GD_OK ? KR_STAGE_FINAL : KR_STAGE_EXIT;
        }
        for (int q = 0; q < MAXF; ++q) S->times[q] = S->t;  // the top level of a chain runs at t
    } else {
        S->stage = KR_STAGE_EXIT;
    }
}

// The weight tables of a level (psi / psi_odd of kde_bandwidth.py:198-229): x and y entries are separate work items, 2 mf 256
// of them over the 512 threads.  A real call for the reason given at kres_tail: inlined, the constants of exp() stayed
// in VGPRs across the level loop and pushed rows of the matrix into scratch, reloaded inside the FMA loop.
template <bool ODD>
__device__ __attribute__((noinline)) void kres_weights(double* wxs, double* wys, const double* logI, const double* times, int Ls,
                                                       int mf, int top, int tid) {
    // at the top level of a chain every form runs at the same time t, and the y exponent of form q is the x exponent of
    // form mf-1-q: the y tables are the x tables in reverse order (the same expression, the same bits) -- half the exp()
    const int items = top ? mf * KR_F : 2 * mf * KR_F;
    for (int e = tid; e < items; e += KR_T) {
        const int isy = top ? 0 : (e & 1), idx = top ? e : (e >> 1), q = idx >> 8, k = idx & 255;
        double v;
        if (!ODD) {
            v = 0;
            if (k > 0) {
                const double I = (double)k * (double)k;
                v = exp(-I * (PISQ * times[q]) + logI[k] * (double)(isy ? Ls - q : q));
            }
        } else {
            const double f = (k <= (KR_F - 1) / 2) ? (double)k : (double)(k - KR_F);
            const int s0 = 1 + 2 * q;
            v = exp(-(f * f) * (4.0 * PISQ * times[q])) * pow(f, (double)(isy ? Ls - s0 : s0));
        }
        if (top) {
            wxs[idx] = v;
            wys[(mf - 1 - q) * KR_F + k] = v;
        } else {
            (isy ? wys : wxs)[idx] = v;
        }
    }
}

// ODD = false: the fixed point and the even functionals on a2.  ODD = true (a second launch, pairs with do_corr only): the
// odd functionals on the power spectrum at the t* and psi_00 the first launch left in `out`.  (One kernel that reloaded
// its registers with the second matrix half-way kept neither in registers: 1200 VGPRs spilled.)
// A level costs three barriers: [weight tables] | [forms -> per-wave partials] | [kres_tail] |.
template <bool ODD>
__global__ void __launch_bounds__(KR_T) k_kopt2d_res(const double* __restrict__ M_all, const KoptPair* __restrict__ pairs,
                                                     double* __restrict__ out) {
    extern __shared__ double lds[];
    KresView V;
    V.wxs = lds;
    V.wys = lds + MAXF * KR_F;
    V.mt = V.wys + MAXF * KR_F;
    V.logI = V.mt + 8 * KR_LDS * KR_F;
    V.sh = reinterpret_cast<KresShared*>(V.logI + KR_F);
    KresShared* S = V.sh;
    const int b = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    double* o = out + (int64_t)b * KOPT_STRIDE;
    if (ODD && !(pairs[b].do_corr && o[7] == (double)GD_OK)) return;  // uniform
    double m[KR_REG][4];
    RPROF_T0;
    kres_load(M_all + (int64_t)(ODD ? pairs[b].pw_index : b) * KR_F * KR_F, m, V.mt, w, l);
    // tables of the constants in the plug-in times, built with the operations (and their order) of kde_bandwidth.py:191-193
    // and :215-219 so that times come out bit for bit as in k_kopt2d; their pow / sqrt run beside the matrix load
    if (!ODD) {
        if (threadIdx.x < 6) S->scale[threadIdx.x] = pow(PI, (double)(2 * threadIdx.x));  // pi^(2L)
        if (threadIdx.x >= 64 && threadIdx.x < 64 + 5 * MAXF) {  // level Ls = 0..4, form a <= Ls
            const int Ls = (threadIdx.x - 64) / MAXF, a = (threadIdx.x - 64) % MAXF;
            if (a <= Ls) {
                const double cst = (1.0 + pow(0.5, (double)(Ls + 1))) / 3.0;
                S->ceven[Ls][a] = -2.0 * cst * k_even(a) * k_even(Ls - a);
            }
        }
        if (threadIdx.x >= 256) V.logI[threadIdx.x - 256] = log((double)(threadIdx.x - 256) * (double)(threadIdx.x - 256));
    } else {
        if (threadIdx.x >= 8 && threadIdx.x < 14) S->scale[threadIdx.x - 2] = pow(2.0 * PI, (double)(2 * (threadIdx.x - 8)));  // (2pi)^L
        if (threadIdx.x >= 128 && threadIdx.x < 138) S->kodd[threadIdx.x - 128] = k_odd(threadIdx.x - 128);
        if (threadIdx.x >= 192 && threadIdx.x < 195) {  // level Ls = 4, 6, 8
            const int Ls = 4 + 2 * (threadIdx.x - 192);
            S->codd[threadIdx.x - 192] = 8.0 * (1.0 - pow(2.0, (double)(-Ls - 1))) / 3.0;
        }
    }
    if (threadIdx.x == 0) {
        const KoptPair P = pairs[b];
        S->N = P.N, S->fallback_t = P.fallback_t, S->do_corr = P.do_corr;
        S->xpre = 0.0, S->xcur = 0.1, S->xblk = 0.0, S->fblk = 0.0, S->spre = 0.0, S->scur = 0.0, S->t_star = 0.0;
        S->fpre = 0.0, S->fcur = 0.0;
        S->status = GD_OK, S->done = 0, S->it = -2;
        S->x = 0.0;
        if (!ODD) {
            S->stage = KR_STAGE_BRENT, S->Ls = 5, S->Lmin = 2, S->t = 0.0;
        } else {
            S->stage = KR_STAGE_ODD, S->Ls = 10, S->Lmin = 4, S->t = o[0];
            S->p00 = o[4];
        }
        for (int q = 0; q < MAXF; ++q) S->times[q] = S->t;
    }
    for (int e = threadIdx.x; e < 6 * MAXF; e += KR_T) (&S->lev[0][0])[e] = 0.0;
#ifdef KOPT_PROFILE
    if (threadIdx.x < 8) S->prof[threadIdx.x] = 0, S->wave_cycles[threadIdx.x] = 0;
#endif
    __syncthreads();
    RPROF_ADD(1);
#pragma unroll 1
    for (;;) {
        const int stage = S->stage, Ls = S->Ls;
        if (stage == KR_STAGE_EXIT) break;
        const int mf = ODD ? Ls / 2 : Ls + 1;  // odd: forms [1+2q, Ls-1-2q]; even: [a, Ls-a]
        kres_weights<ODD>(V.wxs, V.wys, V.logI, S->times, Ls, mf, Ls == (ODD ? 10 : 5), threadIdx.x);
        RPROF_ADD(0);
        __syncthreads();
        RPROF_ADD(5);
        kres_forms(m, V, mf, w, l);
        __syncthreads();
        RPROF_ADD(3);
        if (threadIdx.x < MAXF) kres_tail<ODD>(S, Ls, threadIdx.x);
        RPROF_ADD(7);
        __syncthreads();
        RPROF_ADD(4);
    }
#ifdef KOPT_PROFILE
    if (threadIdx.x < 8 && blockIdx.x == 0 && !ODD) g_prof[threadIdx.x] += S->prof[threadIdx.x], g_wave[threadIdx.x] += S->wave_cycles[threadIdx.x];
#endif
    if (threadIdx.x == 0) {
        if (!ODD) {
            const bool ok = S->status == GD_OK;
            const bool dc = ok && S->do_corr;
            o[0] = ok ? S->t_star : NAN;
            o[1] = ok ? S->lev[2][0] : NAN, o[2] = ok ? S->lev[2][2] : NAN, o[3] = ok ? S->lev[2][1] : NAN;
            o[4] = dc ? S->lev[0][0] : NAN;
            o[5] = NAN, o[6] = NAN;
            o[7] = (double)S->status;
        } else {
            o[5] = S->odd[0][0];  // [1,3]
            o[6] = S->odd[0][1];  // [3,1]
        }
    }
}

// ---- get_h: closed-form bandwidths + the two TNC minimisations of the AMISE (kde_bandwidth.py:234-306) -------------
// The AMISE is a function of ~9 scalars; scipy's TNC (tnc.c + finite-difference gradients) is ported in solvers.hpp and
// pinned against scipy evaluation by evaluation.  One wavefront per pair, lane 0 runs the scalar optimiser (a few
// hundred AMISE evaluations); hundreds of pairs run concurrently on separate SIMDs, so the whole stage costs about one
// pair's serial latency instead of a host process pool.
// kopt: B x KOPT_STRIDE doubles {t*, psi_02, psi_20, psi_11, psi_00, psi_13, psi_31, status, hx, hy, c, get_h status}
__global__ void __launch_bounds__(64) k_get_h(double* __restrict__ kopt, const double* __restrict__ neff,
                                              const double* __restrict__ corr, const int* __restrict__ do_corr, int B) {
    const int b = blockIdx.x;
    // lanes 0..3 run the optimiser in lock step (identical state); inside Tnc::function each of them evaluates one of the
    // n + 1 points of a function-and-gradient call (solvers.hpp); lane 0 writes the result
    if (b >= B || threadIdx.x >= 4) return;
    double* o = kopt + (int64_t)b * KOPT_STRIDE;
    const bool ok = o[7] == (double)GD_OK;
    double psi[6];
    for (int q = 0; q < 6; ++q) psi[q] = o[1 + q];
    if (!ok) {
        if (threadIdx.x == 0) o[8] = o[9] = o[10] = NAN, o[11] = (double)GD_ERR_SOLVER;
        return;
    }
    const gdsolve::GetHResult r = gdsolve::get_h(psi, neff[b], corr[b], do_corr[b] != 0);
    if (threadIdx.x == 0) {
        o[8] = r.hx, o[9] = r.hy, o[10] = r.corr;
        o[11] = r.status ? (double)GD_ERR_BADARG : (double)GD_OK;  // status 1: "bias not positive definite"
    }
}

extern "C" {

int gd_get_h(gd_ctx* ctx, int32_t B, const double* psi, const double* neff, const double* corr, const int32_t* do_corr,
             double* out) {
    GD_REQUIRE(ctx && psi && neff && corr && do_corr && out && B > 0, "bad argument");
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_k = take((int64_t)B * KOPT_STRIDE * 8), o_n = take((int64_t)B * 8), o_c = take((int64_t)B * 8),
                  o_d = take((int64_t)B * 4);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    std::vector<double> hk((size_t)B * KOPT_STRIDE, 0.0);
    for (int b = 0; b < B; ++b) {
        for (int q = 0; q < 6; ++q) hk[(size_t)b * KOPT_STRIDE + 1 + q] = psi[(size_t)b * 6 + q];
        hk[(size_t)b * KOPT_STRIDE + 7] = (double)GD_OK;
    }
    GD_TRY(gd_h2d(ctx, base + o_k, hk.data(), hk.size() * 8));
    GD_TRY(gd_h2d(ctx, base + o_n, neff, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, base + o_c, corr, (size_t)B * 8));
    GD_TRY(gd_h2d(ctx, base + o_d, do_corr, (size_t)B * 4));
    k_get_h<<<B, 64, 0, ctx->stream>>>((double*)(base + o_k), (const double*)(base + o_n), (const double*)(base + o_c),
                                      (const int*)(base + o_d), B);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, hk.data(), base + o_k, hk.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < 4; ++q) out[(size_t)b * 4 + q] = hk[(size_t)b * KOPT_STRIDE + 8 + q];
    return GD_OK;
}

// Stage A of gd_kopt2d (everything up to the functionals), enqueued on ctx's stream: d_rows (device, B x 12) receives
// {t*, psi_02, psi_20, psi_11, psi_00, psi_13, psi_31, status} per pair.
static int kopt_stage_a(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const double* neff, const int32_t* do_corr,
                        const double* fallback_t, double* d_rows) {
    GD_REQUIRE(ctx && d_hist_v && neff && do_corr && fallback_t && d_rows && B > 0, "bad argument");
    GD_REQUIRE(F >= 16 && F <= 1024, "KernelOptimizer2D grid size out of range (16..1024)");
    const double* d_hist = (const double*)d_hist_v;
    const int64_t FF = (int64_t)F * F;
    // DCT route: half-length transforms in LDS (k_dct_pass) where F / 2 is on the transform ladder, else two GEMMs against
    // the cached DCT-II matrix
    FftDev plH;
    const double2 *d_tw = nullptr, *d_ck = nullptr;
    bool dct_fft = F % 2 == 0 && F / 2 <= 512 && F >= 64 && getenv("GDHIP_KOPT_DCT_GEMM") == nullptr &&
                   lds_fft_plan(ctx, F / 2, &plH, nullptr);
    if (dct_fft) {
        FftDev plF;
        dct_fft = lds_fft_plan(ctx, F, &plF, &d_tw);  // (only its table of F twiddles is used)
    }
    if (dct_fft) {
        auto it = ctx->fft_tw.find(-F);  // key -F: the post-twiddles e^{-i pi k / 2F}
        if (it == ctx->fft_tw.end()) {
            std::vector<double2> hck((size_t)F);
            for (int k = 0; k < F; ++k) {
                const long double a = -3.141592653589793238462643383279502884L * (long double)k / (2.0L * (long double)F);
                hck[k] = make_double2((double)cosl(a), (double)sinl(a));
            }
            void* d = nullptr;
            GD_HIP(hipMalloc(&d, (size_t)F * 16));
            if (hipMemcpy(d, hck.data(), (size_t)F * 16, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(d);
                return gd_fail(ctx, GD_ERR_HIP, "upload of the DCT post-twiddles failed");
            }
            it = ctx->fft_tw.emplace(-F, d).first;
        }
        d_ck = (const double2*)it->second;
    }
    double* D = nullptr;
    if (!dct_fft) {
        auto it = ctx->dctmat.find(F);
        if (it == ctx->dctmat.end()) {
            GD_HIP(hipMalloc((void**)&D, (size_t)FF * 8));
            k_dct_matrix<<<(unsigned)((FF + 255) / 256), 256, 0, ctx->stream>>>(F, D);
            GD_KERNEL_CHECK();
            ctx->dctmat[F] = D;
        } else {
            D = it->second;
        }
    }
    std::vector<KoptPair> hp((size_t)B);
    std::vector<int> which;
    for (int b = 0; b < B; ++b) {
        hp[b].N = neff[b];
        hp[b].fallback_t = fallback_t[b];
        hp[b].do_corr = do_corr[b] ? 1 : 0;
        hp[b].pw_index = -1;
        if (do_corr[b]) {
            hp[b].pw_index = (int)which.size();
            which.push_back(b);
        }
    }
    const int nc = (int)which.size();
    const int Fh = F / 2 + 1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    // the per-pair tables sit next to one another: ONE staged upload instead of five pageable copies (each of which
    // would wait for the copy engine behind the previous call's result copies)
    const int64_t o_pairs = take((int64_t)B * sizeof(KoptPair)), o_which = take((int64_t)(nc + 1) * 4);
    const int64_t table_bytes = off;
    const int64_t o_sums = take((int64_t)B * 8), o_E = take((int64_t)B * FF * 8),
                  o_SQ = take((int64_t)B * FF * 8), o_Z = take((int64_t)nc * F * Fh * 16), o_PW = take((int64_t)nc * FF * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    double* d_sums = (double*)(base + o_sums);
    KoptPair* d_pairs = (KoptPair*)(base + o_pairs);
    double* d_out = d_rows;
    int* d_which = (int*)(base + o_which);
    double* d_E = (double*)(base + o_E);
    double* d_SQ = (double*)(base + o_SQ);
    double2* d_Z = (double2*)(base + o_Z);
    double* d_PW = (double*)(base + o_PW);
    {
        std::vector<char> tab((size_t)table_bytes, 0);
        memcpy(tab.data() + o_pairs, hp.data(), (size_t)B * sizeof(KoptPair));
        if (nc) memcpy(tab.data() + o_which, which.data(), (size_t)nc * 4);
        GD_TRY(gd_stage_h2d(ctx, base, tab.data(), (size_t)table_bytes));
    }
    k_hist_sums<<<B, 1024, 0, ctx->stream>>>(d_hist, (int)FF, d_sums);
    GD_KERNEL_CHECK();
    if (dct_fft) {
        const int RPB = F <= 256 ? 16 : (F <= 512 ? 8 : 4);
        const size_t lds = ((size_t)2 * F + (size_t)RPB * F) * 16;
        const dim3 grid((F + RPB - 1) / RPB, B);
        const bool big = F / 2 > 320;
        auto k0 = big ? k_dct_pass<0, true> : k_dct_pass<0, false>;
        auto k1 = big ? k_dct_pass<1, true> : k_dct_pass<1, false>;
        GD_HIP(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GD_HIP(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // E[l][r] = DCT along x of row r (transposed);  SQ[k][l] = (DCT along y of E[l][.] / sum)^2
        k0<<<grid, RPB * FT, lds, ctx->stream>>>(d_hist, F, plH, d_tw, d_ck, nullptr, d_E);
        GD_KERNEL_CHECK();
        k1<<<grid, RPB * FT, lds, ctx->stream>>>(d_E, F, plH, d_tw, d_ck, d_sums, d_SQ);
        GD_KERNEL_CHECK();
    } else {
        const dim3 ggrid((F + 63) / 64, (F + 63) / 64, B);
        // E[l][r] = sum_c D[l][c] X[r][c]      (DCT along axis 1, transposed)
        k_gemm_nt<0><<<ggrid, 256, 0, ctx->stream>>>(D, 0, d_hist, FF, F, d_E, FF, nullptr);
        GD_KERNEL_CHECK();
        // A^T[l][k] = sum_r E[l][r] D[k][r];  SQ[k][l] = (A[k][l] / sum)^2
        k_gemm_nt<1><<<ggrid, 256, 0, ctx->stream>>>(d_E, FF, D, 0, F, d_SQ, FF, d_sums);
        GD_KERNEL_CHECK();
    }
    if (nc > 0) {
        // reuse E as the gathered, normalised input of the FFT (E is dead after the second GEMM)
        k_gather_norm<<<dim3(64, nc), 256, 0, ctx->stream>>>(d_hist, d_which, d_sums, (int)FF, d_E);
        GD_KERNEL_CHECK();
        int rc = gd_fft_r2c_2d(ctx, F, F, nc, d_E, d_Z);
        if (rc) return rc;
        k_power_full<<<dim3(64, nc), 256, 0, ctx->stream>>>(d_Z, F, d_PW);
        GD_KERNEL_CHECK();
    }
    const size_t lds_base = ((size_t)2 * MAXF * F + 8 + 16 + 8 + 16) * 8;
    // row staging area for the bilinear forms: as much of the 160 KB as is left (at most 128 KB), whole 1-KB segments
    size_t tile_bytes = 0;
    if (F % 128 == 0 && getenv("GDHIP_KOPT_NO_LDS_TILE") == nullptr) {
        const size_t room = (size_t)160 * 1024 - 1024 - lds_base;
        size_t cap = (size_t)128 * 1024;
        if (const char* e = getenv("GDHIP_KOPT_TILE_KB")) cap = (size_t)atoi(e) * 1024;  // tuning knob
        tile_bytes = room < cap ? room : cap;
        tile_bytes = tile_bytes / ((size_t)F * 8) * ((size_t)F * 8);  // whole rows even when every segment is needed
    }
    const size_t lds = lds_base + tile_bytes;
    if (F == KR_F && getenv("GDHIP_KOPT_STREAMED") == nullptr) {  // matrix resident on the CU (the switch: A/B tests)
        const size_t lds_res = ((size_t)2 * MAXF * KR_F + (size_t)8 * KR_LDS * KR_F + KR_F) * 8 + sizeof(KresShared);
        GD_HIP(hipFuncSetAttribute((const void*)k_kopt2d_res<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res));
        k_kopt2d_res<false><<<B, KR_T, lds_res, ctx->stream>>>(d_SQ, d_pairs, d_out);
        if (nc > 0) {
            GD_KERNEL_CHECK();
            GD_HIP(hipFuncSetAttribute((const void*)k_kopt2d_res<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res));
            k_kopt2d_res<true><<<B, KR_T, lds_res, ctx->stream>>>(d_PW, d_pairs, d_out);
        }
    } else {
        GD_HIP(hipFuncSetAttribute((const void*)k_kopt2d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_kopt2d<<<B, KT, lds, ctx->stream>>>(d_SQ, d_PW, d_pairs, F, (int)(tile_bytes / 8), d_out);
    }
    GD_KERNEL_CHECK();
#ifdef KOPT_PROFILE
    {
        long long hp[8];
        GD_TRY(gd_stream_sync(ctx));
        GD_HIP(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_prof), sizeof(hp)));
        fprintf(stderr, "kopt block 0 cycles: weights %lld  tile-load %lld  tile-fma %lld  reduce %lld  "
                        "entry-barrier %lld  times/pow %lld | levels (streamed) or barrier after the weights (resident) %lld  kmax sum %lld\n",
                hp[0], hp[1], hp[2], hp[3], hp[4], hp[7], hp[5], hp[6]);
        long long hw[8];
        GD_HIP(hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wave), sizeof(hw)));
        fprintf(stderr, "  forms per wave: %lld %lld %lld %lld %lld %lld %lld %lld\n", hw[0], hw[1], hw[2], hw[3], hw[4], hw[5], hw[6], hw[7]);
        memset(hp, 0, sizeof(hp));
        GD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), hp, sizeof(hp)));
        GD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wave), hp, sizeof(hp)));
    }
#endif
    return GD_OK;
}

// Stage B: KernelOptimizer2D.get_h for the rows stage A left in d_rows, on ctx's stream (which may be another context's
// than stage A's: `after` is then an event recorded behind stage A's kernels); the finished rows come back to `out`.
// The per-pair inputs of get_h (N_eff, correlation, do_corr) sit behind the rows in the same block, uploaded with stage
// A's tables: stage B reads nothing from the host, so its launch never waits for a table to cross PCIe.
#define KOPT_BLOCK_DOUBLES 15  // per pair: 12 result doubles + neff + corr + do_corr (as int in the low half of a double slot)
static int kopt_stage_b(gd_ctx* ctx, hipEvent_t after, int32_t B, double* d_rows, double* out) {
    GD_REQUIRE(ctx && d_rows && out && B > 0, "bad argument");
    if (after) GD_HIP(hipStreamWaitEvent(ctx->stream, after, 0));
    const double* d_neff = d_rows + (int64_t)B * KOPT_STRIDE;
    const double* d_corr = d_neff + B;
    const int* d_dc = (const int*)(d_corr + B);
    k_get_h<<<B, 64, 0, ctx->stream>>>(d_rows, d_neff, d_corr, d_dc, B);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, out, d_rows, (size_t)B * KOPT_STRIDE * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

// upload of get_h's inputs behind the rows (on the stream stage A runs on)
static int kopt_upload_get_h_inputs(gd_ctx* ctx, int32_t B, double* d_rows, const double* neff, const double* corr,
                                    const int32_t* do_corr) {
    std::vector<double> tab((size_t)3 * B, 0.0);
    memcpy(tab.data(), neff, (size_t)B * 8);
    memcpy(tab.data() + B, corr, (size_t)B * 8);
    memcpy(tab.data() + 2 * (size_t)B, do_corr, (size_t)B * 4);
    return gd_h2d(ctx, d_rows + (int64_t)B * KOPT_STRIDE, tab.data(), tab.size() * 8);
}

int gd_kopt2d(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const double* neff, const int32_t* do_corr,
              const double* fallback_t, const double* corr, double* out) {
    GD_REQUIRE(ctx && d_hist_v && neff && do_corr && fallback_t && corr && out && B > 0, "bad argument");
    double* d_rows = (double*)gd_scratch2(ctx, (int64_t)B * KOPT_BLOCK_DOUBLES * 8);
    if (!d_rows) return GD_ERR_NOMEM;
    GD_TRY(kopt_upload_get_h_inputs(ctx, B, d_rows, neff, corr, do_corr));
    GD_TRY(kopt_stage_a(ctx, B, F, d_hist_v, neff, do_corr, fallback_t, d_rows));
    return kopt_stage_b(ctx, nullptr, B, d_rows, out);
}

int gd_kopt2d_enqueue(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr,
                      const double* fallback_t, const double* corr, void* d_rows, int32_t* ticket_out) {
    GD_REQUIRE(ctx && ticket_out && d_rows && neff && corr && do_corr && B > 0, "bad argument");
    GD_TRY(kopt_upload_get_h_inputs(ctx, B, (double*)d_rows, neff, corr, do_corr));
    GD_TRY(kopt_stage_a(ctx, B, F, d_hist, neff, do_corr, fallback_t, (double*)d_rows));
    const int slot = ctx->kopt_ev_next;
    ctx->kopt_ev_next = (slot + 1) % gd_ctx::kKoptEvents;
    if (!ctx->kopt_evs[slot]) GD_HIP(hipEventCreateWithFlags(&ctx->kopt_evs[slot], hipEventDisableTiming));
    GD_HIP(hipEventRecord(ctx->kopt_evs[slot], ctx->stream));
    *ticket_out = slot;
    return GD_OK;
}

int gd_kopt2d_finish(gd_ctx* ctx, gd_ctx* stage_a_ctx, int32_t ticket, int32_t B, void* d_rows, double* out) {
    GD_REQUIRE(ctx && stage_a_ctx && ticket >= 0 && ticket < gd_ctx::kKoptEvents && stage_a_ctx->kopt_evs[ticket], "bad ticket");
    GD_HIP(hipSetDevice(ctx->device));
    return kopt_stage_b(ctx, ctx == stage_a_ctx ? nullptr : stage_a_ctx->kopt_evs[ticket], B, (double*)d_rows, out);
}

}  // extern "C"
