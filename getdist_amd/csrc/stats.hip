// Weighted moments, covariance, sort-free weighted quantiles, autocovariance lags, Gaussian-kernel lag sums.
// All streaming kernels read the SoA columns with 16-byte (double2) loads, coalesced across the wave.
#include "ctx.hpp"

#include <algorithm>
#include <cmath>

#define NBLK_STREAM 1024  // blocks per column for streaming reductions (x 256 threads): >> 256 CUs

// Stream rows [lo,hi) of x (and w) with double2 loads; f(xval, wval) per element.
template <bool HAS_W, class F>
__device__ __forceinline__ void stream_xw(const double* __restrict__ x, const double* __restrict__ w, int64_t lo,
                                          int64_t hi, F f) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t a = (lo + 1) & ~(int64_t)1, b = hi & ~(int64_t)1;  // 16-byte aligned body [a,b)
    if (b <= a) {
        if (gtid == 0)
            for (int64_t i = lo; i < hi; ++i) f(x[i], HAS_W ? w[i] : 1.0);
        return;
    }
    if (gtid == 0) {
        if (lo < a) f(x[lo], HAS_W ? w[lo] : 1.0);
        if (b < hi) f(x[b], HAS_W ? w[b] : 1.0);
    }
    for (int64_t i = a + 2 * gtid; i < b; i += 2 * gsz) {
        const double2 xv = *reinterpret_cast<const double2*>(x + i);
        double2 wv = make_double2(1.0, 1.0);
        if (HAS_W) wv = *reinterpret_cast<const double2*>(w + i);
        f(xv.x, wv.x);
        f(xv.y, wv.y);
    }
}

// Same traversal with four 16-byte loads per array in flight before any element is consumed: for kernels whose
// per-element work (LDS atomics, branches) keeps the compiler from overlapping loop iterations by itself.
template <bool HAS_W, class F>
__device__ __forceinline__ void stream_xw4(const double* __restrict__ x, const double* __restrict__ w, int64_t lo,
                                           int64_t hi, F f) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t a = (lo + 1) & ~(int64_t)1, b = hi & ~(int64_t)1;
    if (b <= a) {
        if (gtid == 0)
            for (int64_t i = lo; i < hi; ++i) f(x[i], HAS_W ? w[i] : 1.0);
        return;
    }
    if (gtid == 0) {
        if (lo < a) f(x[lo], HAS_W ? w[lo] : 1.0);
        if (b < hi) f(x[b], HAS_W ? w[b] : 1.0);
    }
    int64_t i = a + 2 * gtid;
    for (; i + 6 * gsz < b; i += 8 * gsz) {
        double2 xv[4], wv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const double2*>(x + i + 2 * q * gsz);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            wv[q] = HAS_W ? *reinterpret_cast<const double2*>(w + i + 2 * q * gsz) : make_double2(1.0, 1.0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f(xv[q].x, wv[q].x);
            f(xv[q].y, wv[q].y);
        }
    }
    for (; i < b; i += 2 * gsz) {
        const double2 xv = *reinterpret_cast<const double2*>(x + i);
        const double2 wv = HAS_W ? *reinterpret_cast<const double2*>(w + i) : make_double2(1.0, 1.0);
        f(xv.x, wv.x);
        f(xv.y, wv.y);
    }
}

// ---- weight statistics --------------------------------------------------------------------------------
__global__ void k_weight_stats(const double* __restrict__ w, int64_t lo, int64_t hi, double thresh,
                               double* __restrict__ part) {
    __shared__ double red[16];
    double s = 0, s2 = 0, mx = -INFINITY, cnt = 0;
    stream_xw<false>(w, nullptr, lo, hi, [&](double v, double) {
        s += v;
        s2 += v * v;
        mx = fmax(mx, v);
        cnt += (v > thresh) ? 1.0 : 0.0;
    });
    double r0 = block_sum(s, red), r1 = block_max(mx, red), r2 = block_sum(s2, red), r3 = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        double* p = part + (int64_t)blockIdx.x * 4;
        p[0] = r0, p[1] = r1, p[2] = r2, p[3] = r3;
    }
}

// ---- column statistics ----------------------------------------------------------------------------------
// min / max of each column over the rows of [lo,hi) whose condition-column value is < below (cond == nullptr: all
// rows).  grid (blocks, ncols)
template <bool HAS_COND>
__global__ void k_col_minmax(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                             const double* __restrict__ cond, double below, int64_t lo, int64_t hi,
                             double* __restrict__ part) {
    __shared__ double red[16];
    const double* x = cols + (int64_t)colidx[blockIdx.y] * ld;
    double mn = INFINITY, mx = -INFINITY;
    stream_xw<HAS_COND>(x, cond, lo, hi, [&](double v, double c) {
        if (!HAS_COND || c < below) {
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
    });
    double r0 = block_min(mn, red), r1 = block_max(mx, red);
    if (threadIdx.x == 0) {
        double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        p[0] = r0, p[1] = r1;
    }
}

// pass 1: min, max, sum w, sum w x      pass 2: sum w (x-mean)^2
// pass 1 over a GROUP of G columns per block: the weights of a row are loaded once for the G columns (the one-column
// kernel of rounds 1-4 re-read them for every column: 14.2 GB of traffic against 4.04 GB of samples + weights at 100
// columns x 5e6 rows, 2.13 ms per chain; this one 5.5 GB, 0.73 ms: profiles/r05_pmc_c4.json), and a thread has G + 1
// independent 16-byte loads in flight.  grid (NBLK_STREAM, ceil(ncols / G)); the row traversal (stream_xw's), the per-thread
// order of the additions, the wave reduction and the order over the waves (block_sum's) are those of the one-column
// kernel, so the partials -- and every mean downstream -- kept their bits.  Columns past ncols are clamped to the last
// one (loads stay in range, nothing is stored for them).
template <bool HAS_W, int G>
__global__ void __launch_bounds__(256) k_col_pass1g(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                                                    int ncols, const double* __restrict__ w, int64_t lo, int64_t hi,
                                                    double* __restrict__ part) {
    __shared__ double red[4][G][4];  // [wave][column][min, max, sum w, sum w x]
    const double* x[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int slot = min((int)blockIdx.y * G + g, ncols - 1);
        x[g] = cols + (int64_t)(colidx ? colidx[slot] : slot) * ld;
    }
    double mn[G], mx[G], swx[G], sw = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) mn[g] = INFINITY, mx[g] = -INFINITY, swx[g] = 0;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t a = (lo + 1) & ~(int64_t)1, b = hi & ~(int64_t)1;  // 16-byte aligned body [a,b), as stream_xw
    auto one = [&](int64_t i) {
        const double wt = HAS_W ? w[i] : 1.0;
        sw += wt;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double v = x[g][i];
            mn[g] = fmin(mn[g], v), mx[g] = fmax(mx[g], v), swx[g] += wt * v;
        }
    };
    if (b <= a) {
        if (gtid == 0)
            for (int64_t i = lo; i < hi; ++i) one(i);
    } else {
        if (gtid == 0) {
            if (lo < a) one(lo);
            if (b < hi) one(b);
        }
        for (int64_t i = a + 2 * gtid; i < b; i += 2 * gsz) {
            double2 xv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) xv[g] = *reinterpret_cast<const double2*>(x[g] + i);
            double2 wv = make_double2(1.0, 1.0);
            if (HAS_W) wv = *reinterpret_cast<const double2*>(w + i);
            sw += wv.x;
            sw += wv.y;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                mn[g] = fmin(mn[g], xv[g].x), mx[g] = fmax(mx[g], xv[g].x), swx[g] += wv.x * xv[g].x;
                mn[g] = fmin(mn[g], xv[g].y), mx[g] = fmax(mx[g], xv[g].y), swx[g] += wv.y * xv[g].y;
            }
        }
    }
    const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
    const double wsw = wave_sum(sw);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double r0 = wave_min(mn[g]), r1 = wave_max(mx[g]), r3 = wave_sum(swx[g]);
        if (lane == 0) red[wvi][g][0] = r0, red[wvi][g][1] = r1, red[wvi][g][2] = wsw, red[wvi][g][3] = r3;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x, slot = blockIdx.y * G + g;
        if (slot < ncols) {
            double r0 = red[0][g][0], r1 = red[0][g][1], r2 = 0, r3 = 0;
            for (int i = 1; i < 4; ++i) r0 = fmin(r0, red[i][g][0]), r1 = fmax(r1, red[i][g][1]);
            for (int i = 0; i < 4; ++i) r2 += red[i][g][2], r3 += red[i][g][3];
            double* p = part + ((int64_t)slot * gridDim.x + blockIdx.x) * 4;
            p[0] = r0, p[1] = r1, p[2] = r2, p[3] = r3;
        }
    }
}

// one block per column: reduce the pass-1 partials; res[c] = {min, max, sumw, mean}
__global__ void k_col_fin1(const double* __restrict__ part, int nblk, double* __restrict__ res) {
    __shared__ double red[16];
    const double* p = part + (int64_t)blockIdx.x * nblk * 4;
    double mn = INFINITY, mx = -INFINITY, sw = 0, swx = 0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        mn = fmin(mn, p[i * 4 + 0]);
        mx = fmax(mx, p[i * 4 + 1]);
        sw += p[i * 4 + 2];
        swx += p[i * 4 + 3];
    }
    double r0 = block_min(mn, red), r1 = block_max(mx, red), r2 = block_sum(sw, red), r3 = block_sum(swx, red);
    if (threadIdx.x == 0) {
        double* o = res + (int64_t)blockIdx.x * 4;
        o[0] = r0, o[1] = r1, o[2] = r2, o[3] = r3 / r2;
    }
}

template <bool HAS_W>
__global__ void k_col_pass2(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                            const double* __restrict__ w, int64_t lo, int64_t hi, const double* __restrict__ res,
                            double* __restrict__ part) {
    __shared__ double red[16];
    const int c = colidx ? colidx[blockIdx.y] : blockIdx.y;
    const double* x = cols + (int64_t)c * ld;
    const double mean = res[(int64_t)blockIdx.y * 4 + 3];
    double s = 0;
    stream_xw<HAS_W>(x, w, lo, hi, [&](double v, double wt) {
        const double d = v - mean;
        s += wt * (d * d);
    });
    double r = block_sum(s, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// out[c] = {min, max, mean, var}
__global__ void k_col_fin2(const double* __restrict__ part, int nblk, const double* __restrict__ res,
                           double* __restrict__ out) {
    __shared__ double red[16];
    const double* p = part + (int64_t)blockIdx.x * nblk;
    double s = 0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += p[i];
    double r = block_sum(s, red);
    if (threadIdx.x == 0) {
        const double* q = res + (int64_t)blockIdx.x * 4;
        double* o = out + (int64_t)blockIdx.x * 4;
        o[0] = q[0], o[1] = q[1], o[2] = q[3], o[3] = r / q[2];
    }
}

// ---- weighted covariance (two-pass) -----------------------------------------------------------------------
// One block = one 64x64 tile of column pairs over one chunk of rows; 256 threads = 4 waves of 32x32 outputs on the
// fp64 matrix cores.
#define CT 64
#define CRB 64
template <bool HAS_W>
__global__ void __launch_bounds__(256) k_cov_tile(const double* __restrict__ cols, int64_t ld,
                                                  const int32_t* __restrict__ colidx, int m,
                                                  const double* __restrict__ res, const double* __restrict__ w,
                                                  int64_t lo, int64_t hi, int64_t rows_per_chunk,
                                                  const int2* __restrict__ tiles, double* __restrict__ part) {
    __shared__ double sI[CRB][CT + 1];
    __shared__ double sJ[CRB][CT + 1];
    const int2 tl = tiles[blockIdx.y];
    const int ci0 = tl.x * CT, cj0 = tl.y * CT;
    const bool diag = (tl.x == tl.y);
    // 32 x 32 outputs per wave as 2 x 2 v_mfma_f64_16x16x4_f64 tiles (A[i = lane&15][k = lane>>4] = sI[k][i],
    // B[k][j = lane&15] = sJ[k][j], D[row = (lane>>4) + 4 r][col = lane&15]): two LDS reads per lane feed 1024 FMAs
    // per wave and instruction, where the scalar-FMA register tile needed eight reads for sixteen
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32, l15 = lane & 15, lk = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int64_t c_lo = lo + (int64_t)blockIdx.x * rows_per_chunk;
    int64_t c_hi = c_lo + rows_per_chunk;
    if (c_hi > hi) c_hi = hi;
    for (int64_t r0 = c_lo; r0 < c_hi; r0 += CRB) {
        __syncthreads();
        // stage d = x - mean for the I tile (weighted) and the J tile
        for (int e = threadIdx.x; e < CT * CRB; e += 256) {
            const int r = e % CRB, c = e / CRB;
            const int64_t row = r0 + r;
            double vi = 0, vj = 0;
            if (row < c_hi) {
                const double wt = HAS_W ? w[row] : 1.0;
                if (ci0 + c < m) {
                    const int cc = colidx[ci0 + c];
                    vi = (cols[(int64_t)cc * ld + row] - res[(int64_t)(ci0 + c) * 4 + 3]) * wt;
                }
                if (!diag && cj0 + c < m) {
                    const int cc = colidx[cj0 + c];
                    vj = cols[(int64_t)cc * ld + row] - res[(int64_t)(cj0 + c) * 4 + 3];
                } else if (diag && ci0 + c < m) {
                    const int cc = colidx[ci0 + c];
                    vj = cols[(int64_t)cc * ld + row] - res[(int64_t)(ci0 + c) * 4 + 3];
                }
            }
            sI[r][c] = vi;
            sJ[r][c] = vj;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < CRB / 4; ++kk) {
            const int k = kk * 4 + lk;
            const double a0 = sI[k][wi + l15], a1 = sI[k][wi + 16 + l15];
            const double b0 = sJ[k][wj + l15], b1 = sJ[k][wj + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (CT * CT);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(wi + u * 16 + lk + 4 * r) * CT + wj + v * 16 + l15] = acc[u][v][r];
}

// cov[i][j] = sum over chunks / norm, mirrored
__global__ void k_cov_fin(const double* __restrict__ part, int nchunks, const int2* __restrict__ tiles, int m,
                          const double* __restrict__ res, double* __restrict__ cov) {
    const int2 tl = tiles[blockIdx.y];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= CT * CT) return;
    const int i = tl.x * CT + e / CT, j = tl.y * CT + e % CT;
    if (i >= m || j >= m) return;
    if (tl.x == tl.y && j < i) return;  // diagonal tiles: the upper triangle decides, so the result is exactly symmetric
    const double* p = part + (int64_t)blockIdx.y * nchunks * (CT * CT) + e;
    double s = 0;
    for (int c = 0; c < nchunks; ++c) s += p[(int64_t)c * (CT * CT)];
    const double norm = res[2];
    cov[(int64_t)i * m + j] = s / norm;
    cov[(int64_t)j * m + i] = s / norm;
}

// ---- weighted covariance, slab form: every column of a row slab is read from HBM exactly once -------------------------
// A block walks its chunk of rows in slabs of KS rows.  The slab of ALL m columns (d = x - mean, and w d when weighted)
// is staged in LDS once and every upper-triangle 16 x 16 tile pair (ti <= tj) of the covariance is accumulated from it
// by v_mfma_f64_16x16x4_f64; accumulators stay in registers for the whole chunk.
//   LDS layout s[col][KS + 2] (row index fastest): staging writes are contiguous per column, and the MFMA operand reads
//   (lane: col = 16 t + (lane & 15), row = k0 + (lane >> 4)) hit 64 distinct banks per half-wave because the column
//   stride KS + 2 doubles is 4 dwords modulo 64 (MI355X_MICROARCH.md, ds_read_b64 lane groups).
//   * staging: a thread owns two consecutive rows of NQ columns: unconditional 16-byte global loads (column index
//     clamped, the store predicated instead), a ring of DEPTH slabs in flight while the previous slab is multiplied.
//     Ragged ends of a chunk (an odd first row, the last partial slab) go through a guarded slow path once per block;
//   * the waves are an NG x NH arrangement: group g owns P consecutive tile pairs (dead slots alias pair 0 and are not
//     written), row-split h owns KS/4/NH of the slab's k-steps; the loop over (k-step, slot) is straight-line code
//     and sched_group_barrier keeps several operand reads in flight ahead of the matrix pipe;
//   * tile-pair bases are wave-uniform (readfirstlane): one VGPR holds the lane part of every operand address.
// Partials: part[block * T + pair][16 x 16] (the NH row-splits are added inside the block); k_cov_slab_fin sums the blocks.
// Templates: MCAP = column capacity (multiple of 16), NW = NG * NH waves per block, P tile pairs per wave, KS rows per
// slab, DEPTH slabs of global loads in flight.
// ONE PASS (OP, round 6): `res` holds a provisional SHIFT s per column (k_col_shift: the mean of a few hundred strided rows)
// instead of the mean, and the slab carries one more column, all ones (w when weighted on the A side): the same tile
// products then also deliver sum w (x - s) per column and sum w, from which k_cov_onepass_fin forms
//   mean = s + delta,  delta = sum w (x - s) / sum w,   cov_ij = sum w (x_i - s_i)(x_j - s_j) / sum w  -  delta_i delta_j
// -- the textbook shifted form, as accurate as the two-pass one because delta^2 is ~1e-3 of the variance (the cancellation
// costs 1e-3 ulp, not digits); min / max ride along in the staging registers.  The means pass (k_col_pass1g: one read of the
// whole sample block) is gone.
template <bool HAS_W, int MCAP, int NW, int NH, int P, int KS, int DEPTH, bool OP = false>
__global__ void __launch_bounds__(NW * 64, (OP && MCAP == 64) ? 4 : 1) k_cov_slab2(const double* __restrict__ cols, int64_t ld,
                                                       const int32_t* __restrict__ colidx, int m,
                                                       const double* __restrict__ res, const double* __restrict__ w,
                                                       int64_t lo, int64_t hi, int64_t rows_per_chunk,
                                                       double* __restrict__ part, double* __restrict__ part_mm = nullptr) {
    constexpr int NT = NW * 64, KSP = KS + 2, TPC = KS / 2, CPP = NT / TPC, NG = NW / NH;
    constexpr int NQ = (MCAP + CPP - 1) / CPP, KPW = KS / 4 / NH;
    static_assert(NW % NH == 0 && (KS / 4) % NH == 0 && NT % TPC == 0, "wave arrangement");
    extern __shared__ double lds[];
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int me = OP ? m + 1 : m;  // columns of the slab (OP: the ones column behind the data)
    const int nt = (me + 15) / 16, mc = nt * 16, T = nt * (nt + 1) / 2;
    double* sB = lds;                            // d
    double* sA = HAS_W ? lds + mc * KSP : lds;   // w d (the same array for unit weights)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wv % NG, h = wv / NG;
    int offA[P], offB[P];  // wave-uniform element offsets of the slots' tiles
    f64x4 acc[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        int pq = g * P + q, a = 0, len = nt;
        if (pq >= T) pq = 0;
        while (pq >= len) {
            pq -= len;
            ++a;
            --len;
        }
        offA[q] = a * 16 * KSP;
        offB[q] = (a + pq) * 16 * KSP;
        acc[q] = (f64x4){0.0, 0.0, 0.0, 0.0};
    }
    const int lbase = l15 * KSP + lk + h * KPW * 4;  // lane part: column l15 of the tile, row lk of the wave's k-steps
    for (int e = tid; e < (HAS_W ? 2 : 1) * mc * KSP; e += NT) lds[e] = 0.0;  // padding columns / rows stay zero
    // staging: this thread owns rows 2 rp, 2 rp + 1 of columns c0 + CPP q
    const int rp = tid % TPC, c0 = tid / TPC;
    const double* src[NQ];
    double mean_r[OP ? 1 : NQ];
    double cmn[OP ? NQ : 1], cmx[OP ? NQ : 1];
    // OP keeps the columns' shifts in LDS behind the slab (read back per slab): its running extrema need the registers, and
    // with more than 128 of them a CU holds one block instead of two
    double* sM = lds + (HAS_W ? 2 : 1) * mc * KSP;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = c0 + CPP * q, cc = c < m ? c : m - 1;
        src[q] = cols + (int64_t)colidx[cc] * ld;
        if (OP) {
            cmn[q] = INFINITY, cmx[q] = -INFINITY;
            if (rp == 0 && c < mc) sM[c] = res[(int64_t)cc * 4 + 3];
        } else {
            mean_r[q] = res[(int64_t)cc * 4 + 3];
        }
    }
    auto mean_of = [&](int q) -> double { return OP ? sM[(c0 + CPP * q) < mc ? c0 + CPP * q : mc - 1] : mean_r[q]; };
    const int64_t c_lo = lo + (int64_t)blockIdx.x * rows_per_chunk;
    int64_t c_hi = c_lo + rows_per_chunk;
    if (c_hi > hi) c_hi = hi;

    // the (k-step, slot) sequence as straight-line code with the NEXT product's two operand reads issued before each
    // MFMA (two register pairs alternate): the matrix pipe never waits for a read it has just asked for, and the
    // operand registers stay at eight whatever P is
    // the (k-step, slot) products of one k-step as straight-line code; the schedule groups ask for the NEXT product's
    // two operand reads to be issued before each MFMA, so the matrix pipe never waits for a read it has just requested
    constexpr int UNRK = KPW * P <= 28 ? KPW : 2;
    auto multiply = [&]() {
#pragma unroll UNRK
        for (int kk = 0; kk < KPW; ++kk) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const double a = sA[offA[q] + lbase + kk * 4], b = sB[offB[q] + lbase + kk * 4];
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
            }
            if (P >= 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS reads of the first two products
#pragma unroll
                for (int q = 0; q < P - 2; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA ...
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // ... then the reads of the product after next
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
    };
    // guarded slab: rows [rb, re), re - rb <= KS, any alignment (once or twice per block)
    auto slab_guarded = [&](int64_t rb, int64_t re) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = c0 + CPP * q;
            if (c < m) {
                const double* x = cols + (int64_t)colidx[c] * ld;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int64_t row = rb + 2 * rp + e;
                    const bool in = row < re;
                    const double mq = mean_of(q);
                    const double xv = in ? x[row] : mq;
                    if (OP && in) cmn[q] = fmin(cmn[q], xv), cmx[q] = fmax(cmx[q], xv);
                    const double d = in ? xv - mq : 0.0;
                    sB[c * KSP + 2 * rp + e] = d;
                    if (HAS_W) sA[c * KSP + 2 * rp + e] = in ? d * w[row] : 0.0;
                }
            } else if (OP && c == m) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int64_t row = rb + 2 * rp + e;
                    const bool in = row < re;
                    sB[c * KSP + 2 * rp + e] = in ? 1.0 : 0.0;
                    if (HAS_W) sA[c * KSP + 2 * rp + e] = in ? w[row] : 0.0;
                }
            }
        }
        __syncthreads();
        multiply();
    };
    if (c_lo < c_hi) {
        const int64_t a_lo = (c_lo + 1) & ~(int64_t)1;  // first 16-byte aligned row of the chunk
        const int64_t nfull = c_hi > a_lo ? (c_hi - a_lo) / KS : 0;
        if (a_lo > c_lo) slab_guarded(c_lo, a_lo < c_hi ? a_lo : c_hi);
        // DEPTH slabs of loads in flight per lane (a register ring): narrow matrices have so little matrix work per
        // slab that one slab ahead does not cover the HBM latency
        double2 pre[DEPTH][NQ], wpre[DEPTH];
        auto fetch = [&](int d, int64_t slab) {
            const int64_t v = a_lo + slab * KS + 2 * rp;
#pragma unroll
            for (int q = 0; q < NQ; ++q) pre[d][q] = gload_d2(src[q] + v);
            wpre[d] = HAS_W ? gload_d2(w + v) : make_double2(1.0, 1.0);
        };
        const int64_t nring = nfull / DEPTH * DEPTH;  // slabs taken by the ring: whole rounds only, no inner branches
        if (nring > 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                fetch(d, d);
                __builtin_amdgcn_sched_barrier(0);  // in ring order
            }
        }
        for (int64_t s = 0; s < nring; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                __syncthreads();  // the previous slab's operand reads are done
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = c0 + CPP * q;
                    if (c < m) {
                        if (OP) {
                            cmn[q] = fmin(cmn[q], fmin(pre[d][q].x, pre[d][q].y));
                            cmx[q] = fmax(cmx[q], fmax(pre[d][q].x, pre[d][q].y));
                        }
                        const double mq = mean_of(q);
                        const double2 dd = make_double2(pre[d][q].x - mq, pre[d][q].y - mq);
                        *reinterpret_cast<double2*>(&sB[c * KSP + 2 * rp]) = dd;
                        if (HAS_W)
                            *reinterpret_cast<double2*>(&sA[c * KSP + 2 * rp]) = make_double2(dd.x * wpre[d].x, dd.y * wpre[d].y);
                    } else if (OP && c == m) {
                        *reinterpret_cast<double2*>(&sB[c * KSP + 2 * rp]) = make_double2(1.0, 1.0);
                        if (HAS_W) *reinterpret_cast<double2*>(&sA[c * KSP + 2 * rp]) = wpre[d];
                    }
                }
                __syncthreads();
                // in flight while DEPTH slabs are multiplied; unconditional (past the end: a harmless re-read of the
                // last slab) so that the number of loads in flight is the same on every path and the wait before the
                // LDS stores can leave the other ring slots' loads outstanding
                fetch(d, s + d + DEPTH < nring ? s + d + DEPTH : nring - 1);
                multiply();
            }
        }
        // what the ring left: fewer than DEPTH whole slabs and the ragged end
        for (int64_t t_lo = a_lo + nring * KS; t_lo < c_hi; t_lo += KS) slab_guarded(t_lo, t_lo + KS < c_hi ? t_lo + KS : c_hi);
    }
    if (OP) {
        // the TPC threads that staged the same columns (consecutive lanes of one wave) combine their extrema
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double a = cmn[q], b = cmx[q];
#pragma unroll
            for (int o = 1; o < TPC; o <<= 1) {
                a = fmin(a, __shfl_xor(a, o, WAVE));
                b = fmax(b, __shfl_xor(b, o, WAVE));
            }
            const int c = c0 + CPP * q;
            if (rp == 0 && c < m) {
                part_mm[((int64_t)blockIdx.x * m + c) * 2] = a;
                part_mm[((int64_t)blockIdx.x * m + c) * 2 + 1] = b;
            }
        }
    }
    if (NH == 1) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int pq = g * P + q;
            if (pq < T) {
                double* p = part + ((int64_t)blockIdx.x * T + pq) * 256;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) p[(lk + 4 * rg) * 16 + l15] = acc[q][rg];
            }
        }
    } else {
        // the NH row-splits of a tile pair are added up here, in the fixed order h = 0, 1, ..., through LDS (the staging
        // area is free now): one partial per (block, pair) leaves the kernel, and the sum is the same in every run
        __syncthreads();
        double* red = lds;  // NG * P * 256 doubles (the launch sizes the dynamic LDS for it)
#pragma unroll 1
        for (int hh = 0; hh < NH; ++hh) {
            if (h == hh) {
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int idx = (g * P + q) * 256 + (lk + 4 * rg) * 16 + l15;
                        red[idx] = hh == 0 ? acc[q][rg] : red[idx] + acc[q][rg];
                    }
            }
            __syncthreads();
        }
        for (int e = tid; e < NG * P * 256; e += NT) {
            const int pq = e >> 8;
            if (pq < T) part[((int64_t)blockIdx.x * T + pq) * 256 + (e & 255)] = red[e];
        }
    }
}

// cov[i][j] = sum over blocks / norm, mirrored; grid (pairs), 1024 threads: four interleaved partial sums per element
// (four times the loads in flight), combined in a fixed order
__global__ void __launch_bounds__(1024) k_cov_slab_fin(const double* __restrict__ part, int nblocks, int m,
                                                       const double* __restrict__ res, double* __restrict__ cov) {
    __shared__ double sh[4][256];
    const int nt = (m + 15) / 16, T = nt * (nt + 1) / 2;
    int pq = blockIdx.x, a = 0, len = nt;
    while (pq >= len) {
        pq -= len;
        ++a;
        --len;
    }
    const int e = threadIdx.x & 255, lane4 = threadIdx.x >> 8;
    const double* p = part + (int64_t)blockIdx.x * 256 + e;
    double s0 = 0, s1 = 0;
    int b = lane4;
    for (; b + 4 < nblocks; b += 8) {
        s0 += p[(int64_t)b * T * 256];
        s1 += p[(int64_t)(b + 4) * T * 256];
    }
    if (b < nblocks) s0 += p[(int64_t)b * T * 256];
    sh[lane4][e] = s0 + s1;
    __syncthreads();
    if (lane4 != 0) return;
    const double v = ((sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e])) / res[2];
    const int i = a * 16 + e / 16, j = (a + pq) * 16 + e % 16;
    if (i >= m || j >= m) return;
    if (pq == 0 && j < i) return;  // diagonal tiles: the upper triangle decides, so the result is exactly symmetric
    cov[(int64_t)i * m + j] = v;
    cov[(int64_t)j * m + i] = v;
}

// ---- one-pass form (k_cov_slab2<..., OP = true>) ------------------------------------------------------------------------
// provisional shift per column: the plain mean of up to 256 rows strided over [lo, hi) (any value near the mean will do:
// it only has to make |mean - shift| small against the spread).  One block per column.
__global__ void __launch_bounds__(256) k_col_shift(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                                                   int64_t lo, int64_t hi, double* __restrict__ res) {
    __shared__ double red[16];
    const double* x = cols + (int64_t)colidx[blockIdx.x] * ld;
    const int64_t rows = hi - lo;
    const int64_t k = rows < 256 ? rows : 256;
    double v = 0;
    if ((int64_t)threadIdx.x < k) v = x[lo + (int64_t)threadIdx.x * (rows / k)];
    const double sum = block_sum(v, red);
    if (threadIdx.x == 0) {
        double s = sum / (double)k;
        if (!(fabs(s) < 1e300)) s = 0.0;  // (non-finite samples: the statistics are NaN either way)
        res[(int64_t)blockIdx.x * 4 + 3] = s;
    }
}

// S[i][j] = sum over the blocks' partial tiles, both triangles; mc1 = 16 * tiles per side.  grid (T), 1024 threads, the
// summation order of k_cov_slab_fin.
__global__ void __launch_bounds__(1024) k_cov_slab_sum(const double* __restrict__ part, int nblocks, int nt, double* __restrict__ S) {
    __shared__ double sh[4][256];
    const int T = nt * (nt + 1) / 2, mc1 = nt * 16;
    int pq = blockIdx.x, a = 0, len = nt;
    while (pq >= len) {
        pq -= len;
        ++a;
        --len;
    }
    const int e = threadIdx.x & 255, lane4 = threadIdx.x >> 8;
    const double* p = part + (int64_t)blockIdx.x * 256 + e;
    double s0 = 0, s1 = 0;
    int b = lane4;
    for (; b + 4 < nblocks; b += 8) {
        s0 += p[(int64_t)b * T * 256];
        s1 += p[(int64_t)(b + 4) * T * 256];
    }
    if (b < nblocks) s0 += p[(int64_t)b * T * 256];
    sh[lane4][e] = s0 + s1;
    __syncthreads();
    if (lane4 != 0) return;
    const double v = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
    const int i = a * 16 + e / 16, j = (a + pq) * 16 + e % 16;
    if (pq == 0 && j < i) return;  // diagonal tiles: the upper triangle decides (exact symmetry)
    S[(int64_t)i * mc1 + j] = v;
    S[(int64_t)j * mc1 + i] = v;
}

// block c: the column's extrema over the blocks, its mean, and row c of the covariance (upper part, mirrored).
// res[c] = {min, max, sum w, mean} as k_col_fin1 leaves it.
__global__ void __launch_bounds__(256) k_cov_onepass_fin(const double* __restrict__ S, int mc1, int m, const double* __restrict__ part_mm,
                                                         int nblocks, double* __restrict__ res, double* __restrict__ cov) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    double mn = INFINITY, mx = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        mn = fmin(mn, part_mm[((int64_t)b * m + c) * 2]);
        mx = fmax(mx, part_mm[((int64_t)b * m + c) * 2 + 1]);
    }
    const double r0 = block_min(mn, red), r1 = block_max(mx, red);
    const double norm = S[(int64_t)m * mc1 + m];
    const double dc = S[(int64_t)c * mc1 + m] / norm;
    for (int j = c + threadIdx.x; j < m; j += 256) {
        const double dj = S[(int64_t)j * mc1 + m] / norm;
        const double v = S[(int64_t)c * mc1 + j] / norm - dc * dj;
        cov[(int64_t)c * m + j] = v;
        cov[(int64_t)j * m + c] = v;
    }
    __syncthreads();  // (every thread has read the shift before thread 0 replaces it by the mean)
    if (threadIdx.x == 0) {
        const double shift = res[(int64_t)c * 4 + 3];
        res[(int64_t)c * 4 + 0] = r0, res[(int64_t)c * 4 + 1] = r1, res[(int64_t)c * 4 + 2] = norm, res[(int64_t)c * 4 + 3] = shift + dc;
    }
}

// ---- sort-free weighted quantiles: MSB radix select, 8 bits per pass ---------------------------------------
#define QK_MAX 16
struct QState {  // per column
    unsigned long long prefix[QK_MAX];  // selected key bits so far (right-aligned)
    double cum_below[QK_MAX];           // weight strictly below the selected prefix bucket
    double target[QK_MAX];
    int slot[QK_MAX];                   // target -> unique-prefix slot
    unsigned long long uprefix[QK_MAX];
    int nuniq;
    int k;
    unsigned long long bloom;  // 64-bit Bloom mask over the unique prefixes (one hash)
};

__device__ __forceinline__ int prefix_hash6(unsigned long long top) {
    unsigned int t = (unsigned int)(top ^ (top >> 32));
    t ^= t >> 16;
    t ^= t >> 8;
    return (int)((t ^ (t >> 4)) & 63u);
}

__device__ __forceinline__ unsigned long long f64_key(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// All lanes of the wave that target the same bin contribute through one LDS atomic.  In pass 0 the digit is the
// sign + top exponent bits, so whole waves hit 1-4 bins and plain atomics would serialise 64-deep on one address.
template <bool HAS_W>
__device__ __forceinline__ void wave_agg_add(double* h, int bin, double wt, bool valid) {
    // must be called by all 64 lanes of the wave (lanes without data pass valid = false)
    unsigned long long todo = __ballot(valid);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lb = __builtin_amdgcn_readlane(bin, leader);
        const unsigned long long same = __ballot(bin == lb) & todo;
        double v;
        if (HAS_W) {
            v = ((same >> lane) & 1ull) ? wt : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
        } else {
            v = (double)__popcll(same);  // unit weights: the group's weight is its population count
        }
        if (lane == leader) atomicAdd(&h[lb], v);
        todo &= ~same;
    }
}

template <bool HAS_W>
__global__ void k_qsel_pass(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                            const double* __restrict__ w, int64_t lo, int64_t hi, int pass,
                            const QState* __restrict__ st, double* __restrict__ ghist) {
    __shared__ double h[QK_MAX * 256];
    __shared__ unsigned long long upre[QK_MAX];
    __shared__ int nu;
    const int c = blockIdx.y;
    const double* x = cols + (int64_t)colidx[c] * ld;
    if (threadIdx.x == 0) nu = (pass == 0) ? 1 : st[c].nuniq;
    if (threadIdx.x < QK_MAX) upre[threadIdx.x] = st[c].uprefix[threadIdx.x];
    __syncthreads();
    const int nuniq = nu;
    for (int i = threadIdx.x; i < nuniq * 256; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    if (pass == 0) {
        // uniform trip count per wave so that every lane takes part in the wave-level aggregation
        const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i0 = lo + (int64_t)blockIdx.x * blockDim.x; i0 < hi; i0 += 4 * gsz) {
            double v[4], wt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = i0 + q * gsz + threadIdx.x;
                const bool ok = i < hi;
                v[q] = ok ? x[i] : 0.0;
                wt[q] = ok ? (HAS_W ? w[i] : 1.0) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wave_agg_add<HAS_W>(h, (int)(f64_key(v[q]) >> 56), wt[q], i0 + q * gsz + threadIdx.x < hi);
        }
    } else {
        const unsigned long long bloom = st[c].bloom;
        unsigned long long pre[QK_MAX];  // uniform per block: the compiler keeps these in scalar registers
#pragma unroll
        for (int u = 0; u < QK_MAX; ++u) pre[u] = st[c].uprefix[u];
        stream_xw4<HAS_W>(x, w, lo, hi, [&](double v, double wt) {
            const unsigned long long key = f64_key(v);
            const unsigned long long top = key >> (shift + 8);
            if ((bloom >> prefix_hash6(top)) & 1ull) {
                int slot = -1;
#pragma unroll
                for (int u = 0; u < QK_MAX; ++u)
                    if (u < nuniq && top == pre[u]) slot = u;
                if (slot >= 0) atomicAdd(&h[slot * 256 + (int)((key >> shift) & 255ull)], wt);
            }
        });
    }
    __syncthreads();
    double* g = ghist + (int64_t)c * QK_MAX * 256;
    for (int i = threadIdx.x; i < nuniq * 256; i += blockDim.x) {
        const double v = h[i];
        if (v != 0) unsafeAtomicAdd(&g[i], v);
    }
}

// one block per column: consume the histograms, extend each target's prefix by 8 bits, dedupe prefixes.
__global__ void k_qsel_scan(QState* __restrict__ st, double* __restrict__ ghist, int ncols, int pass) {
    const int c = blockIdx.x;
    QState& s = st[c];
    double* g = ghist + (int64_t)c * QK_MAX * 256;
    const int t = threadIdx.x;
    if (t < s.k) {
        const double* h = g + (pass == 0 ? 0 : s.slot[t]) * 256;
        double cum = s.cum_below[t];
        int pick = -1, last_nonempty = -1;
        double cum_at_last = cum;
        for (int b = 0; b < 256; ++b) {
            const double hv = h[b];
            if (hv != 0) {
                last_nonempty = b;
                cum_at_last = cum;
            }
            if (cum + hv >= s.target[t] && hv != 0) {
                pick = b;
                break;
            }
            cum += hv;
        }
        if (pick < 0) {  // target beyond the total weight: the reference clamps to the last element
            pick = last_nonempty < 0 ? 0 : last_nonempty;
            cum = cum_at_last;
        }
        s.prefix[t] = (s.prefix[t] << 8) | (unsigned long long)pick;
        s.cum_below[t] = cum;
    }
    __syncthreads();
    if (t == 0) {  // unique prefixes for the next pass
        int nu = 0;
        for (int q = 0; q < s.k; ++q) {
            int f = -1;
            for (int u = 0; u < nu; ++u)
                if (s.uprefix[u] == s.prefix[q]) f = u;
            if (f < 0) {
                f = nu;
                s.uprefix[nu++] = s.prefix[q];
            }
            s.slot[q] = f;
        }
        s.nuniq = nu;
        unsigned long long bl = 0;
        for (int u = 0; u < nu; ++u) bl |= 1ull << prefix_hash6(s.uprefix[u]);
        s.bloom = bl;
    }
    __syncthreads();
    for (int i = t; i < QK_MAX * 256; i += blockDim.x) g[i] = 0;
}

__global__ void k_qsel_out(const QState* __restrict__ st, int ncols, int k, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols * k) return;
    out[i] = key_f64(st[i / k].prefix[i % k]);
}

// ---- early finish of the radix select ------------------------------------------------------------------------
// After P passes (8P key bits known) the selected buckets of continuous data hold a handful of rows (N = 1e7
// Gaussian columns: <~1000 after 24 bits, <~10 after 32).  Instead of five more streaming passes, ONE pass collects
// the (key, weight) pairs of every live bucket and a per-column block sorts each short list and walks its cumulative
// weight exactly like np.cumsum on the argsorted column does (chains.py:807-811, 834-838).  A bucket that overflows
// the list (heavily tied data) sets a flag and the caller falls back to the remaining radix passes.
#define QCAP 4096
template <bool HAS_W>
__global__ void k_qsel_collect(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                               const double* __restrict__ w, int64_t lo, int64_t hi, int passes_done,
                               const QState* __restrict__ st, unsigned long long* __restrict__ lkeys,
                               double* __restrict__ lw, int* __restrict__ counts) {
    const int c = blockIdx.y;
    const double* x = cols + (int64_t)colidx[c] * ld;
    const int nuniq = st[c].nuniq;
    const int shift = 64 - 8 * passes_done;
    const unsigned long long bloom = st[c].bloom;
    unsigned long long pre[QK_MAX];
#pragma unroll
    for (int u = 0; u < QK_MAX; ++u) pre[u] = st[c].uprefix[u];
    stream_xw4<HAS_W>(x, w, lo, hi, [&](double v, double wt) {
        const unsigned long long key = f64_key(v);
        const unsigned long long top = key >> shift;
        if ((bloom >> prefix_hash6(top)) & 1ull) {
            int slot = -1;
#pragma unroll
            for (int u = 0; u < QK_MAX; ++u)
                if (u < nuniq && top == pre[u]) slot = u;
            if (slot >= 0) {
                const int pos = atomicAdd(&counts[c * QK_MAX + slot], 1);
                if (pos < QCAP) {
                    const int64_t o = ((int64_t)c * QK_MAX + slot) * QCAP + pos;
                    lkeys[o] = key;
                    lw[o] = wt;
                }
            }
        }
    });
}

// one block per (column, live bucket) -- grid (ncols, QK_MAX): sort the bucket's list (bitonic, LDS) and pick the row of
// every target that lives in it.  (One block per column sorting its up to 11 lists in turn took 1.1 ms of a 45-ms step.)
__global__ void __launch_bounds__(1024) k_qsel_finish(const QState* __restrict__ st, const unsigned long long* __restrict__ lkeys,
                                                      const double* __restrict__ lw, const int* __restrict__ counts, int k,
                                                      int passes_done, double* __restrict__ out, int* __restrict__ overflow,
                                                      int unit) {
    __shared__ unsigned long long sk[QCAP];
    __shared__ double sw[QCAP];
    const int c = blockIdx.x;
    const QState& s = st[c];
    if ((int)blockIdx.y >= s.nuniq) return;
    if (counts[c * QK_MAX + blockIdx.y] > QCAP) {
        if (threadIdx.x == 0) atomicOr(overflow, 1);
        return;
    }
    {
        const int u = blockIdx.y;
        const int n = counts[c * QK_MAX + u];
        int m = 1;
        while (m < n) m <<= 1;
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const int64_t o = ((int64_t)c * QK_MAX + u) * QCAP + i;
            sk[i] = (i < n) ? lkeys[o] : ~0ull;
            sw[i] = (i < n) ? lw[o] : 0.0;
        }
        __syncthreads();
        for (int size = 2; size <= m; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = threadIdx.x; i < m; i += blockDim.x) {
                    const int j = i ^ stride;
                    if (j > i) {
                        const bool up = (i & size) == 0;
                        const unsigned long long a = sk[i], b = sk[j];
                        if ((a > b) == up) {
                            sk[i] = b, sk[j] = a;
                            const double t = sw[i];
                            sw[i] = sw[j], sw[j] = t;
                        }
                    }
                }
                __syncthreads();
            }
        // every target that lives in this bucket walks the sorted list (thread q <-> target q)
        if (threadIdx.x < k && s.slot[threadIdx.x] == u) {
            const int q = threadIdx.x;
            double cum = s.cum_below[q];
            int pick = -1, last = -1;
            if (unit && n > 0) {
                // unit weights (round 6): after i rows the walk below holds cum_below + i exactly (row counts), so the row it
                // stops at is known without walking -- a guess from the difference, settled with the walk's own comparison
                // on its own operands.  (One thread reading ~1000 list entries one LDS latency apart was what this kernel's
                // 0.25 ms consisted of; it sits on the chain in front of a triangle's first grids.)
                const double tq = s.target[q];
                int g = (int)fmin(fmax(ceil(tq - cum) - 1.0, 0.0), (double)(n - 1));
                while (g > 0 && (cum + (double)(g - 1)) + 1.0 >= tq) --g;
                while (g < n - 1 && !((cum + (double)g) + 1.0 >= tq)) ++g;
                pick = g;  // (no row reaches the target: the walk's clamp to the last row, g = n - 1)
            }
            for (int i = 0; i < n && pick < 0; ++i) {
                const double wv = sw[i];
                if (wv != 0) {
                    last = i;
                    if (cum + wv >= s.target[q]) {
                        pick = i;
                        break;
                    }
                }
                cum += wv;
            }
            if (pick < 0) pick = last < 0 ? 0 : last;  // beyond the total weight: clamp to the last row
            out[(int64_t)c * k + q] = key_f64(n > 0 ? sk[pick] : (s.prefix[q] << (64 - 8 * passes_done)));
        }
    }
}

// ---- linear-bucket select: two reads of the columns instead of four -----------------------------------------------
// When the column's minimum and maximum are known (the base statistics have them), ONE counting pass over
// b = (int)((x - min) * scale) -- monotone in x, so buckets partition the sort order exactly like radix digits do --
// with 32768 (unit weights, u32 counters) or 16384 (fp64 weight sums) buckets held in LDS narrows every target to a
// bucket of a few hundred to a few thousand rows; the collect pass and k_qsel_finish (sorted walk of the cumulative
// weight, chains.py:807-838) are those of the radix path.  The radix path's first digit is sign + exponent, which
// separates nothing for data of one magnitude: that is why it needs three passes before the collect.
#define QLIN_NB_U 32768
#define QLIN_NB_W 16384
struct QLin {  // per column, next to its QState
    double mn, scale;
    int nb, pad;
    int ubucket[QK_MAX];
};
__device__ __forceinline__ int qlin_bucket(double v, double mn, double scale, int nb) {
    int b = (int)((v - mn) * scale);
    b = b < 0 ? 0 : b;
    return b < nb ? b : nb - 1;
}

template <bool HAS_W>
__global__ void __launch_bounds__(1024) k_qlin_count(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                                                     const double* __restrict__ w, int64_t lo, int64_t hi,
                                                     const QLin* __restrict__ ql, void* __restrict__ part, double wscale) {
    extern __shared__ double qsh[];
    const int c = blockIdx.y;
    const double* x = cols + (int64_t)colidx[c] * ld;
    const double mn = ql[c].mn, scale = ql[c].scale;
    constexpr int nb = HAS_W ? QLIN_NB_W : QLIN_NB_U;
    unsigned int* hu = reinterpret_cast<unsigned int*>(qsh);
    unsigned long long* hq = reinterpret_cast<unsigned long long*>(qsh);
    for (int i = threadIdx.x; i < nb; i += 1024) {
        if (HAS_W)
            hq[i] = 0ull;
        else
            hu[i] = 0u;
    }
    __syncthreads();
    // weights are added as 64-bit integers, round(w * 2^k) with 2^k * (sum of all weights) < 2^62: integer sums are the
    // same in any order, so the bucket totals -- and the sample every target selects -- repeat from run to run, which
    // fp64 LDS atomics did not give for real weights (and ds_add_u64 issues at twice the rate of ds_add_f64)
    auto add = [&](double v, double wt) {
        const int b = qlin_bucket(v, mn, scale, nb);
        if (HAS_W)
            atomicAdd(&hq[b], __double2ull_rn(wt * wscale));
        else
            atomicAdd(&hu[b], 1u);
    };
    {
        // the 128-KB table leaves one block per CU: eight 16-byte loads per lane in flight (the 4 of stream_xw4 left
        // the pass at 2.9 TB/s), ragged ends by the generic walker
        const int64_t gtid = (int64_t)blockIdx.x * 1024 + threadIdx.x, gsz = (int64_t)gridDim.x * 1024;
        const int64_t a = (lo + 1) & ~(int64_t)1, b = hi & ~(int64_t)1;
        constexpr int U = HAS_W ? 4 : 8;
        int64_t i = a + 2 * gtid;
        for (; i + 2 * (U - 1) * gsz < b; i += 2 * U * gsz) {
            double2 xv[U], wv[U];
#pragma unroll
            for (int q = 0; q < U; ++q) xv[q] = gload_d2(x + i + 2 * q * gsz);
#pragma unroll
            for (int q = 0; q < U; ++q) wv[q] = HAS_W ? gload_d2(w + i + 2 * q * gsz) : make_double2(1.0, 1.0);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                add(xv[q].x, wv[q].x);
                add(xv[q].y, wv[q].y);
            }
        }
        for (; i < b; i += 2 * gsz) {
            const double2 xv = gload_d2(x + i);
            const double2 wv = HAS_W ? gload_d2(w + i) : make_double2(1.0, 1.0);
            add(xv.x, wv.x);
            add(xv.y, wv.y);
        }
        if (gtid == 0) {
            if (b <= a) {
                for (int64_t r = lo; r < hi; ++r) add(x[r], HAS_W ? w[r] : 1.0);
            } else {
                if (lo < a) add(x[lo], HAS_W ? w[lo] : 1.0);
                if (b < hi) add(x[b], HAS_W ? w[b] : 1.0);
            }
        }
    }
    __syncthreads();
    if (HAS_W) {
        unsigned long long* p = (unsigned long long*)part + ((int64_t)c * gridDim.x + blockIdx.x) * nb;
        for (int i = threadIdx.x; i < nb; i += 1024) p[i] = hq[i];
    } else {
        unsigned int* p = (unsigned int*)part + ((int64_t)c * gridDim.x + blockIdx.x) * nb;
        for (int i = threadIdx.x; i < nb; i += 1024) p[i] = hu[i];
    }
}

// bucket totals: the blocks' partial tables added in block order; grid (nb / 256, ncols)
template <bool HAS_W>
__global__ void __launch_bounds__(256) k_qlin_reduce(const void* __restrict__ part, int nblk, double* __restrict__ tot,
                                                      double inv_wscale) {
    constexpr int nb = HAS_W ? QLIN_NB_W : QLIN_NB_U;
    const int c = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
    double v = 0;
    if (HAS_W) {
        const unsigned long long* p = (const unsigned long long*)part + (int64_t)c * nblk * nb + b;
        unsigned long long u = 0;
        for (int k = 0; k < nblk; ++k) u += p[(int64_t)k * nb];
        v = (double)u * inv_wscale;  // a power of two: the only rounding is the conversion of the exact integer total
    } else {
        const unsigned int* p = (const unsigned int*)part + (int64_t)c * nblk * nb + b;
        unsigned long long u = 0;
        for (int k = 0; k < nblk; ++k) u += p[(int64_t)k * nb];
        v = (double)u;
    }
    tot[(int64_t)c * nb + b] = v;
}

// per column, from the bucket totals: every target's bucket -- the first non-empty one at which the cumulative weight
// reaches the target -- and the weight below it; fills the QState the collect / finish kernels read.  One block of
// 1024 threads per column; thread t owns PER consecutive buckets.
template <bool HAS_W>
__global__ void __launch_bounds__(1024) k_qlin_scan(QState* __restrict__ st, QLin* __restrict__ ql, const double* __restrict__ totals) {
    constexpr int nb = HAS_W ? QLIN_NB_W : QLIN_NB_U, PER = nb / 1024;
    __shared__ double tsum[1024];
    __shared__ int pick[QK_MAX];
    __shared__ double cumb[QK_MAX];
    __shared__ int lastne;
    const int c = blockIdx.x, t = threadIdx.x;
    QState& s = st[c];
    double tot[PER];
    double mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const double v = totals[(int64_t)c * nb + t * PER + j];
        tot[j] = v;
        mine += v;
    }
    tsum[t] = mine;
    if (t < QK_MAX) pick[t] = 0x7fffffff;
    if (t == 0) lastne = -1;
    __syncthreads();
    // weight of every bucket before this thread's first: exclusive prefix of the thread sums in three fixed steps
    // (lane prefix by shuffles, the 16 wave totals serially, then the add) -- the same association in every run
    double below;
    {
        const int lane = t & 63, wv = t >> 6;
        double incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += up;
        }
        __shared__ double wtot[16];
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        double base = 0;
        for (int k = 0; k < wv; ++k) base += wtot[k];
        below = base + (incl - mine);
    }
    int my_last = -1;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (tot[j] != 0) my_last = t * PER + j;
    if (my_last >= 0) atomicMax(&lastne, my_last);
    for (int q = 0; q < s.k; ++q) {
        const double target = s.target[q];
        double cum = below;
        int found = -1;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const double hv = tot[j];
            if (found < 0 && hv != 0 && cum + hv >= target) found = t * PER + j;
            cum += hv;
        }
        if (found >= 0) atomicMin(&pick[q], found);
    }
    __syncthreads();
    for (int q = 0; q < s.k; ++q) {
        int p = pick[q];
        if (p == 0x7fffffff) p = lastne < 0 ? 0 : lastne;  // target beyond the total weight: the last row (chains.py:836)
        if (p / PER == t) {
            double cum = below;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (t * PER + j < p) cum += tot[j];
            cumb[q] = cum;
            pick[q] = p;
        }
    }
    __syncthreads();
    if (t == 0) {
        int nu = 0;
        for (int q = 0; q < s.k; ++q) {
            s.cum_below[q] = cumb[q];
            s.prefix[q] = 0;
            int f = -1;
            for (int u = 0; u < nu; ++u)
                if (ql[c].ubucket[u] == pick[q]) f = u;
            if (f < 0) {
                f = nu;
                ql[c].ubucket[nu++] = pick[q];
            }
            s.slot[q] = f;
        }
        s.nuniq = nu;
    }
}

// Rows that fall in a live bucket are staged per block (LDS) and appended with ONE global atomic per (block, bucket): a
// column's ~10^4 hits used to take one returning atomic each on 11 counters of one cache line, which serialised in L2 and
// was the whole duration of the collect pass (0.35 ms for 6 columns, 0.45 ms for 50).  Lists stay sets: the finish sorts.
#define QHIT_CAP 2048
template <bool HAS_W>
struct QHits {
    unsigned long long key[QHIT_CAP];
    double wt[HAS_W ? QHIT_CAP : 1];
    unsigned char slot[QHIT_CAP];
    int n, cnt[QK_MAX], base[QK_MAX];
    __device__ void init() {  // followed by a block barrier of the caller
        if (threadIdx.x == 0) n = 0;
        if (threadIdx.x < QK_MAX) cnt[threadIdx.x] = 0;
    }
    __device__ void hit(int c, int sl, unsigned long long k, double w, unsigned long long* __restrict__ lkeys,
                        double* __restrict__ lw, int* __restrict__ counts) {
        const int p = atomicAdd(&n, 1);
        if (p < QHIT_CAP) {
            key[p] = k;
            if (HAS_W) wt[p] = w;
            slot[p] = (unsigned char)sl;
        } else {  // more hits in one block than the stage holds (rows of a bucket clustered in the chain): straight to the list
            const int pos = atomicAdd(&counts[c * QK_MAX + sl], 1);
            if (pos < QCAP) {
                const int64_t o = ((int64_t)c * QK_MAX + sl) * QCAP + pos;
                lkeys[o] = k;
                lw[o] = HAS_W ? w : 1.0;
            }
        }
    }
    __device__ void flush(int c, unsigned long long* __restrict__ lkeys, double* __restrict__ lw, int* __restrict__ counts) {
        __syncthreads();
        const int m = n < QHIT_CAP ? n : QHIT_CAP;
        constexpr int E = QHIT_CAP / 512;
        int q[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int e = threadIdx.x + j * blockDim.x;
            q[j] = e < m ? atomicAdd(&cnt[slot[e]], 1) : 0;
        }
        __syncthreads();
        if (threadIdx.x < QK_MAX && cnt[threadIdx.x] > 0) base[threadIdx.x] = atomicAdd(&counts[c * QK_MAX + threadIdx.x], cnt[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int e = threadIdx.x + j * blockDim.x;
            if (e < m) {
                const int sl = slot[e], pos = base[sl] + q[j];
                if (pos < QCAP) {
                    const int64_t o = ((int64_t)c * QK_MAX + sl) * QCAP + pos;
                    lkeys[o] = key[e];
                    lw[o] = HAS_W ? wt[e] : 1.0;
                }
            }
        }
    }
};

// the (key, weight) pairs of every live bucket, appended to that bucket's list
template <bool HAS_W>
__global__ void __launch_bounds__(512) k_qlin_collect(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                               const double* __restrict__ w, int64_t lo, int64_t hi, const QState* __restrict__ st,
                               const QLin* __restrict__ ql, unsigned long long* __restrict__ lkeys, double* __restrict__ lw,
                               int* __restrict__ counts) {
    constexpr int nb = HAS_W ? QLIN_NB_W : QLIN_NB_U;
    __shared__ unsigned int live[nb / 32];  // one bit per bucket
    __shared__ int ub[QK_MAX];
    __shared__ QHits<HAS_W> hits;
    const int c = blockIdx.y;
    const double* x = cols + (int64_t)colidx[c] * ld;
    const int nuniq = st[c].nuniq;
    const double mn = ql[c].mn, scale = ql[c].scale;
    for (int i = threadIdx.x; i < nb / 32; i += blockDim.x) live[i] = 0u;
    if (threadIdx.x < QK_MAX) ub[threadIdx.x] = threadIdx.x < nuniq ? ql[c].ubucket[threadIdx.x] : -1;
    hits.init();
    __syncthreads();
    if ((int)threadIdx.x < nuniq) atomicOr(&live[ub[threadIdx.x] >> 5], 1u << (ub[threadIdx.x] & 31));
    __syncthreads();
    stream_xw4<HAS_W>(x, w, lo, hi, [&](double v, double wt) {
        const int b = qlin_bucket(v, mn, scale, nb);
        if ((live[b >> 5] >> (b & 31)) & 1u) {
            int slot = 0;
            for (int u = 1; u < nuniq; ++u)
                if (ub[u] == b) slot = u;
            hits.hit(c, slot, f64_key(v), wt, lkeys, lw, counts);
        }
    });
    hits.flush(c, lkeys, lw, counts);
}

// ---- the counting pass over WHOLE columns, fused (round 6) --------------------------------------------------------------
// One read of a column serves three consumers that all need nothing but pass-1 results (min / max / mean):
//   * the bucket counts of the linear quantile select (as k_qlin_count),
//   * the column's BUCKET COLUMN: the 16-bit bucket index of every sample, stored for the passes that follow (collect,
//     pre-binning: ctx.hpp BucketCols) -- 2 bytes per sample written here save 8 bytes per sample read there, twice,
//   * (PROBE) the first QP_LAGS autocovariance lag sums sum_i d_i d_{i+l}, d = (x - mean) w, of getCorrelationLength's first
//     probe (chains.py:423-466; k_autocov computes the same sums from a read of its own).
// Rows are walked in tiles of QP_TILE = 2 x 1024 consecutive rows (a thread owns two consecutive rows of a tile), U tiles of
// loads in flight per lane.  The lag products need no block barrier: a wave's 128 consecutive d values go through a
// wave-private LDS strip, the 8 rows behind the strip (the next wave's / next tile's first rows) come from a load of their
// own (one cache line the neighbour fetches anyway).  The table leaves one block per CU, as for k_qlin_count.
#define QP_TILE 2048
#define QP_LAGS 8
template <bool HAS_W, bool PROBE>
__global__ void __launch_bounds__(1024) k_qlin_count_tiles(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                                                           const double* __restrict__ w, int64_t N, const QLin* __restrict__ ql,
                                                           void* __restrict__ part, double wscale, unsigned short* __restrict__ bq,
                                                           const double* __restrict__ means, double* __restrict__ probe_part) {
    extern __shared__ double qsh[];
    constexpr int nb = HAS_W ? QLIN_NB_W : QLIN_NB_U;
    constexpr int STRIP = 128 + QP_LAGS;  // doubles per wave
    const int c = blockIdx.y, col = colidx[c];
    const double* x = cols + (int64_t)col * ld;
    unsigned short* bqc = bq ? bq + (int64_t)col * ld : nullptr;
    const double mn = ql[c].mn, scale = ql[c].scale;
    unsigned int* hu = reinterpret_cast<unsigned int*>(qsh);
    unsigned long long* hq = reinterpret_cast<unsigned long long*>(qsh);
    double* strip = qsh + (size_t)nb * (HAS_W ? 8 : 4) / 8 + (threadIdx.x >> 6) * STRIP;  // behind the table
    for (int i = threadIdx.x; i < nb; i += 1024) {
        if (HAS_W)
            hq[i] = 0ull;
        else
            hu[i] = 0u;
    }
    __syncthreads();
    const double mean = PROBE ? means[c] : 0.0;
    double acc[QP_LAGS];
#pragma unroll
    for (int l = 0; l < QP_LAGS; ++l) acc[l] = 0;
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (N + QP_TILE - 1) / QP_TILE;
    // (the probe's strip values and accumulators take registers: four tiles in flight keep the kernel free of spills)
    constexpr int U = (HAS_W || PROBE) ? 4 : 8;
    const int64_t G = gridDim.x;
    for (int64_t t = blockIdx.x; t < ntiles; t += U * G) {
        double2 xv[U], wv[U];
        double hx[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t tq = t + q * G;
            // tiles past the end: the loads stay in range (clamped, nothing of them is used)
            const int64_t base = (tq < ntiles ? tq : ntiles - 1) * QP_TILE;
            int64_t r = base + 2 * threadIdx.x;
            if (r > ld - 2) r = ld - 2;
            xv[q] = gload_d2(x + r);
            wv[q] = HAS_W ? gload_d2(w + r) : make_double2(1.0, 1.0);
            if (PROBE) {
                int64_t rh = base + (int64_t)((threadIdx.x >> 6) + 1) * 128 + (lane & (QP_LAGS - 1));
                const bool in = tq < ntiles && rh < N;
                if (rh > ld - 1) rh = ld - 1;
                const double v = x[rh];
                const double ww = HAS_W ? w[rh] : 1.0;
                hx[q] = in ? (v - mean) * ww : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t tq = t + q * G;
            const int64_t row0 = tq < ntiles ? tq * QP_TILE + 2 * threadIdx.x : N;  // (>= N: both rows masked)
            const bool m0 = row0 < N, m1 = row0 + 1 < N;
            const int b0 = qlin_bucket(xv[q].x, mn, scale, nb), b1 = qlin_bucket(xv[q].y, mn, scale, nb);
            if (HAS_W) {
                if (m0) atomicAdd(&hq[b0], __double2ull_rn(wv[q].x * wscale));
                if (m1) atomicAdd(&hq[b1], __double2ull_rn(wv[q].y * wscale));
            } else {
                if (m0) atomicAdd(&hu[b0], 1u);
                if (m1) atomicAdd(&hu[b1], 1u);
            }
            if (bqc && m0) *reinterpret_cast<unsigned int*>(bqc + row0) = (unsigned)b0 | ((unsigned)b1 << 16);
            if (PROBE) {
                const double d0 = m0 ? (xv[q].x - mean) * wv[q].x : 0.0, d1 = m1 ? (xv[q].y - mean) * wv[q].y : 0.0;
                __builtin_amdgcn_wave_barrier();  // the previous tile's strip reads are done (DS operations of a wave are in order)
                strip[2 * lane] = d0;
                strip[2 * lane + 1] = d1;
                if (lane < QP_LAGS) strip[128 + lane] = hx[q];
                // (no fence: a fence would also wait for the U tiles of global loads in flight; the reads below may alias the
                // stores above, so the compiler keeps their order, and the LDS executes a wave's operations in order)
                __builtin_amdgcn_wave_barrier();
                double v[QP_LAGS + 1];
#pragma unroll
                for (int l = 0; l <= QP_LAGS; ++l) v[l] = strip[2 * lane + l];
#pragma unroll
                for (int l = 0; l < QP_LAGS; ++l) acc[l] = fma(d1, v[l + 1], fma(d0, v[l], acc[l]));
            }
        }
    }
    __syncthreads();
    if (HAS_W) {
        unsigned long long* p = (unsigned long long*)part + ((int64_t)c * gridDim.x + blockIdx.x) * nb;
        for (int i = threadIdx.x; i < nb; i += 1024) p[i] = hq[i];
    } else {
        unsigned int* p = (unsigned int*)part + ((int64_t)c * gridDim.x + blockIdx.x) * nb;
        for (int i = threadIdx.x; i < nb; i += 1024) p[i] = hu[i];
    }
    if (PROBE) {
        __syncthreads();
        double* red = qsh;  // the table has been flushed
        double* p = probe_part + ((int64_t)c * gridDim.x + blockIdx.x) * QP_LAGS;
#pragma unroll
        for (int l = 0; l < QP_LAGS; ++l) {
            const double r = block_sum(acc[l], red);
            if (threadIdx.x == 0) p[l] = r;
        }
    }
}

// the collect pass from the bucket column: 2 bytes per sample instead of 8; a sample is re-read (with its weight) only when
// its bucket is live -- about 11 buckets of 32768.  Same lists, same order-independent finish (k_qsel_finish sorts them).
template <bool HAS_W>
__global__ void __launch_bounds__(512) k_qlin_collect_bq(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                                                         const double* __restrict__ w, int64_t N, const QState* __restrict__ st,
                                                         const QLin* __restrict__ ql, const unsigned short* __restrict__ bq,
                                                         unsigned long long* __restrict__ lkeys, double* __restrict__ lw,
                                                         int* __restrict__ counts) {
    constexpr int nb = HAS_W ? QLIN_NB_W : QLIN_NB_U;
    __shared__ unsigned int live[nb / 32];  // one bit per bucket
    __shared__ int ub[QK_MAX];
    __shared__ QHits<HAS_W> hits;
    const int c = blockIdx.y, col = colidx[c];
    const double* x = cols + (int64_t)col * ld;
    const unsigned short* bqc = bq + (int64_t)col * ld;
    const int nuniq = st[c].nuniq;
    for (int i = threadIdx.x; i < nb / 32; i += blockDim.x) live[i] = 0u;
    if (threadIdx.x < QK_MAX) ub[threadIdx.x] = threadIdx.x < nuniq ? ql[c].ubucket[threadIdx.x] : -1;
    hits.init();
    __syncthreads();
    if ((int)threadIdx.x < nuniq) atomicOr(&live[ub[threadIdx.x] >> 5], 1u << (ub[threadIdx.x] & 31));
    __syncthreads();
    auto hit = [&](int64_t row, int b) {
        int slot = 0;
        for (int u = 1; u < nuniq; ++u)
            if (ub[u] == b) slot = u;
        hits.hit(c, slot, f64_key(x[row]), HAS_W ? w[row] : 1.0, lkeys, lw, counts);
    };
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t n8 = (N + 7) / 8;  // 16-byte groups; the column's storage is padded to a multiple of 512 rows
    constexpr int U = 4;
    int64_t i = gtid;
    for (; i + (U - 1) * gsz < n8; i += U * gsz) {
        uint4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = gload_u4(bqc + 8 * (i + q * gsz));
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const unsigned int wd[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int b = (wd[e >> 1] >> (16 * (e & 1))) & (nb - 1);  // (rows past N hold whatever the padding held)
                const int64_t row = 8 * (i + q * gsz) + e;
                if (((live[b >> 5] >> (b & 31)) & 1u) && row < N) hit(row, b);
            }
        }
    }
    for (; i < n8; i += gsz) {
        const uint4 v = gload_u4(bqc + 8 * i);
        const unsigned int wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = (wd[e >> 1] >> (16 * (e & 1))) & (nb - 1);
            const int64_t row = 8 * i + e;
            if (((live[b >> 5] >> (b & 31)) & 1u) && row < N) hit(row, b);
        }
    }
    hits.flush(c, lkeys, lw, counts);
}

// ---- autocovariance lag sums: out[l] = sum_i d_i d_{i+k0+l},  d = (x-mean)*w ---------------------------------
#define AL 32     // lags per launch
#define AT 2048   // rows per tile
template <bool HAS_W, int NL>
__global__ void __launch_bounds__(256) k_autocov(const double* __restrict__ cols, int64_t ld,
                                                 const int32_t* __restrict__ colidx, const double* __restrict__ w_all,
                                                 int64_t row0, int64_t N, const double* __restrict__ means, int64_t k0,
                                                 double* __restrict__ part) {
    __shared__ double sB[AT + NL];
    __shared__ double red[16];
    const double* x = cols + (int64_t)colidx[blockIdx.y] * ld + row0;  // rows [row0, row0+N) of the column
    const double* w = HAS_W ? w_all + row0 : nullptr;
    const double mean = means[blockIdx.y];
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = 0;
    static_assert(AT % 256 == 0 && NL <= 256, "tile shape");
    constexpr int Q = AT / 256;
    for (int64_t t0 = (int64_t)blockIdx.x * AT; t0 < N; t0 += (int64_t)gridDim.x * AT) {
        // the whole tile is requested before anything is consumed: Q independent loads in flight per lane
        double v[Q], vw[Q], hv = 0, hw = 1;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t r = t0 + k0 + threadIdx.x + 256 * q;
            v[q] = (r < N) ? x[r] : mean;
            vw[q] = (HAS_W && r < N) ? w[r] : 1.0;
        }
        if (threadIdx.x < NL) {
            const int64_t r = t0 + k0 + AT + threadIdx.x;
            hv = (r < N) ? x[r] : mean;
            hw = (HAS_W && r < N) ? w[r] : 1.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) sB[threadIdx.x + 256 * q] = (v[q] - mean) * vw[q];
        if (threadIdx.x < NL) sB[AT + threadIdx.x] = (hv - mean) * hw;
        __syncthreads();
        if (k0 == 0) {  // the leading factor is the tile itself
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int e = threadIdx.x + 256 * q;
                const double a = sB[e];  // rows beyond N were stored as 0
#pragma unroll
                for (int l = 0; l < NL; ++l) acc[l] = fma(a, sB[e + l], acc[l]);
            }
        } else {
            for (int e = threadIdx.x; e < AT; e += 256) {
                const int64_t r = t0 + e;
                if (r < N) {
                    const double a = (x[r] - mean) * (HAS_W ? w[r] : 1.0);
#pragma unroll
                    for (int l = 0; l < NL; ++l) acc[l] = fma(a, sB[e + l], acc[l]);
                }
            }
        }
    }
    double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NL;
#pragma unroll
    for (int l = 0; l < NL; ++l) {  // unrolled: acc[] must stay in registers
        const double r = block_sum(acc[l], red);
        if (threadIdx.x == 0) p[l] = r;
    }
}

// out[c][l] = sum_b part[c][b][l];  grid (AL, ncols)
__global__ void k_sum_partials_batched(const double* __restrict__ part, int nblk, int stride, double* __restrict__ out) {
    __shared__ double red[16];
    const double* p = part + (int64_t)blockIdx.y * nblk * stride;
    double s = 0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += p[(int64_t)i * stride + blockIdx.x];
    const double r = block_sum(s, red);
    if (threadIdx.x == 0) out[(int64_t)blockIdx.y * stride + blockIdx.x] = r;
}

// ---- Gaussian-kernel lag sums: out[l] = sum_i exp(-(x_i - x_{i+k})^2 * c) w_i w_{i+k} -----------------------
// exp(-t) for t >= 0 (the only argument range of these sums), ~2e-16 relative: t = (64 q + j) ln2/64 - r with
// |r| <= ln2/128, exp(-t) = 2^-q' * T[j'] * exp(r); T = 2^(j/64) from a 64-entry LDS table, exp(r) by a degree-5
// polynomial (next term 3.5e-17).  About half the instructions of the library exp, which these kernels are bound by.
__device__ __forceinline__ double exp_neg(double t, const double* __restrict__ tab) {
    t = __builtin_fmin(t, 708.0);  // beyond: below 1e-307 anyway, and the int conversion below would wrap for huge t
    const double n = __builtin_rint(t * 92.332482616893656756);  // 64 / ln 2
    // ln2/64 split so that n * hi is exact for n < 2^17
    double r = __builtin_fma(n, 1.0830424696223417413e-02, -t);
    r = __builtin_fma(n, 2.5728046223276688287e-14, r);
    double p = __builtin_fma(r, 8.3333333333333332177e-03, 4.1666666666666664354e-02);
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int m = -(int)n;  // exp(-t) = 2^(m/64) exp(r)
    return __builtin_ldexp(tab[m & 63] * p, m >> 6);
}

#define KDE_LAG_MAX 8
// All NL lags of a column in one read: grid (blocks, columns), rows in chunks of 256.  Chunks below min(N - k) need no
// bounds handling at all (straight-line code: NL loads in flight, NL interleaved exponentials); the others skip a lag
// whose range has ended for the whole chunk (with lags near N/2 that is most of the second half) and mask the rest.
template <bool HAS_W, int NL>
__global__ void __launch_bounds__(256) k_kde_lag_multi(const double* __restrict__ cols, int64_t ld,
                                                       const int32_t* __restrict__ colidx, const double* __restrict__ w,
                                                       int64_t N, const double* __restrict__ cvals,
                                                       const int64_t* __restrict__ lags, double* __restrict__ part) {
    __shared__ double red[16];
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64));
    __syncthreads();
    const double* x = cols + (int64_t)colidx[blockIdx.y] * ld;
    const double c = cvals[blockIdx.y];
    int64_t k[NL], M[NL];
    int64_t Mmin = N;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        k[l] = lags[l];
        M[l] = N - k[l];
        Mmin = M[l] < Mmin ? M[l] : Mmin;
    }
    double s[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) s[l] = 0;
    const int64_t nA = Mmin / 256, nT = (N + 255) / 256;
    for (int64_t ch = blockIdx.x; ch < nA; ch += gridDim.x) {
        const int64_t i = ch * 256 + threadIdx.x;
        const double xi = x[i];
        const double wi = HAS_W ? w[i] : 1.0;
        double xk[NL], wk[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            xk[l] = (x + k[l])[i];
            wk[l] = HAS_W ? (w + k[l])[i] : 1.0;
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const double d = xi - xk[l];
            double e = exp_neg((d * d) * c, tab);
            if (HAS_W) e = e * wi * wk[l];
            s[l] += e;
        }
    }
    for (int64_t ch = nA + blockIdx.x; ch < nT; ch += gridDim.x) {
        const int64_t i0 = ch * 256, i = i0 + threadIdx.x;
        const double xi = i < N ? x[i] : 0.0;
        const double wi = (HAS_W && i < N) ? w[i] : 1.0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (i0 < M[l]) {  // uniform: this lag still has rows in the chunk
                const bool valid = i < M[l];
                const int64_t r = valid ? i + k[l] : N - 1;
                const double d = xi - x[r];
                double e = exp_neg((d * d) * c, tab);
                if (HAS_W) e = e * wi * w[r];
                s[l] += valid ? e : 0.0;
            }
        }
    }
    double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * KDE_LAG_MAX;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const double r = block_sum(s[l], red);
        if (threadIdx.x == 0) p[l] = r;
    }
}

// The same sums with every sample read from HBM once, for the lag sets the N_eff estimate asks for (mcsamples.py:
// NF lags K0 + delta near N/2 and NN short ones).  The kernel above reads the second half of a column twice: as the far
// partners of the first half and again as the x_i of the short lags.  Here only rows i < R = N - K0 are walked and each
// thread also takes the short-lag terms of row j = i + K0, whose lines its far loads have just brought in (rows
// [0, R) as x_i and rows [R, N) as x_j partition the column when 2 K0 <= N; for odd N row i = 0 has no second role).
// lags[]: the NF far lags, then the NN short ones; sums come out in that order.
template <bool HAS_W, int NF, int NN>
__global__ void __launch_bounds__(256) k_kde_lag_folded(const double* __restrict__ cols, int64_t ld,
                                                        const int32_t* __restrict__ colidx, const double* __restrict__ w,
                                                        int64_t N, const double* __restrict__ cvals,
                                                        const int64_t* __restrict__ lags, double* __restrict__ part) {
    __shared__ double red[16];
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64));
    __syncthreads();
    const double* x = cols + (int64_t)colidx[blockIdx.y] * ld;
    const double c = cvals[blockIdx.y];
    int64_t kf[NF], kn[NN];
    int64_t K0 = N, kmax = 0, nmax = 0;
#pragma unroll
    for (int l = 0; l < NF; ++l) {
        kf[l] = lags[l];
        K0 = kf[l] < K0 ? kf[l] : K0;
        kmax = kf[l] > kmax ? kf[l] : kmax;
    }
#pragma unroll
    for (int l = 0; l < NN; ++l) {
        kn[l] = lags[NF + l];
        nmax = kn[l] > nmax ? kn[l] : nmax;
    }
    const int64_t R = N - K0, J0 = N - 2 * K0;  // rows walked; first row with a second role
    double s[NF + NN];
#pragma unroll
    for (int l = 0; l < NF + NN; ++l) s[l] = 0;
    const double* xj = x + K0;
    const double* wj = HAS_W ? w + K0 : nullptr;
    // chunks [cA, cB): every term of every thread exists
    const int64_t nT = (R + 255) / 256, cA = (J0 + 255) / 256;
    int64_t cB = (N - kmax < R - nmax ? N - kmax : R - nmax) / 256;
    if (cB < cA) cB = cA;
    for (int64_t ch = cA + blockIdx.x; ch < cB; ch += gridDim.x) {
        const int64_t i = ch * 256 + threadIdx.x;
        const double xi = x[i], xs = xj[i];
        const double wi = HAS_W ? w[i] : 1.0, ws = HAS_W ? wj[i] : 1.0;
        double xk[NF], wk[NF], xa[NN], wa[NN], xb[NN], wb[NN];
#pragma unroll
        for (int l = 0; l < NF; ++l) {
            xk[l] = (x + kf[l])[i];
            wk[l] = HAS_W ? (w + kf[l])[i] : 1.0;
        }
#pragma unroll
        for (int l = 0; l < NN; ++l) {
            xa[l] = (x + kn[l])[i];
            xb[l] = (xj + kn[l])[i];
            wa[l] = HAS_W ? (w + kn[l])[i] : 1.0;
            wb[l] = HAS_W ? (wj + kn[l])[i] : 1.0;
        }
#pragma unroll
        for (int l = 0; l < NF; ++l) {
            const double d = xi - xk[l];
            double e = exp_neg((d * d) * c, tab);
            if (HAS_W) e = e * wi * wk[l];
            s[l] += e;
        }
#pragma unroll
        for (int l = 0; l < NN; ++l) {
            const double d = xi - xa[l], d2 = xs - xb[l];
            double e = exp_neg((d * d) * c, tab), e2 = exp_neg((d2 * d2) * c, tab);
            if (HAS_W) {
                e = e * wi * wa[l];
                e2 = e2 * ws * wb[l];
            }
            s[NF + l] += e + e2;
        }
    }
    for (int64_t ch = blockIdx.x; ch < nT; ch += gridDim.x) {
        if (ch >= cA && ch < cB) continue;
        const int64_t i = ch * 256 + threadIdx.x;
        const bool row = i < R;
        const double xi = row ? x[i] : 0.0;
        const double wi = (HAS_W && row) ? w[i] : 1.0;
#pragma unroll
        for (int l = 0; l < NF; ++l) {
            const bool valid = i < N - kf[l];
            const int64_t r = valid ? i + kf[l] : N - 1;
            const double d = xi - x[r];
            double e = exp_neg((d * d) * c, tab);
            if (HAS_W) e = e * wi * w[r];
            s[l] += valid ? e : 0.0;
        }
        const bool second = row && i >= J0;
        const int64_t j = second ? i + K0 : N - 1;
        const double xs = x[j];
        const double ws = HAS_W ? w[j] : 1.0;
#pragma unroll
        for (int l = 0; l < NN; ++l) {
            {
                const bool valid = row && i + kn[l] < N;
                const int64_t r = valid ? i + kn[l] : N - 1;
                const double d = xi - x[r];
                double e = exp_neg((d * d) * c, tab);
                if (HAS_W) e = e * wi * w[r];
                s[NF + l] += valid ? e : 0.0;
            }
            {
                const bool valid = second && j + kn[l] < N;
                const int64_t r = valid ? j + kn[l] : N - 1;
                const double d = xs - x[r];
                double e = exp_neg((d * d) * c, tab);
                if (HAS_W) e = e * ws * w[r];
                s[NF + l] += valid ? e : 0.0;
            }
        }
    }
    double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * KDE_LAG_MAX;
#pragma unroll
    for (int l = 0; l < NF + NN; ++l) {
        const double r = block_sum(s[l], red);
        if (threadIdx.x == 0) p[l] = r;
    }
}

template <bool HAS_W>
__global__ void k_kde_lag(const double* __restrict__ cols, int64_t ld, const int32_t* __restrict__ colidx,
                          const double* __restrict__ w, int64_t N, const double* __restrict__ cvals,
                          const int64_t* __restrict__ lags, double* __restrict__ part) {
    __shared__ double red[16];
    const double* x = cols + (int64_t)colidx[blockIdx.z] * ld;
    const double c = cvals[blockIdx.z];
    const int64_t k = lags[blockIdx.y];
    const int64_t M = N - k;
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = x[i] - x[i + k];
        double e = exp(-(d * d) * c);
        if (HAS_W) e = e * w[i] * w[i + k];
        s += e;
    }
    const double r = block_sum(s, red);
    if (threadIdx.x == 0) part[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = r;
}

// 2D variant (chains.py:576-635): sum_i exp(-(d^T Kinv d)/4) w_i w_{i+k}, d = (x_i - x_{i+k}, y_i - y_{i+k})
template <bool HAS_W>
__global__ void k_kde_lag_2d(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                             int64_t N, double k00, double k01s, double k11, const int64_t* __restrict__ lags,
                             double* __restrict__ part) {
    __shared__ double red[16];
    const int64_t k = lags[blockIdx.y];
    const int64_t M = N - k;
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        const double dx = x[i] - x[i + k], dy = y[i] - y[i + k];
        const double diff2 = dx * (k00 * dx) + dx * (k01s * dy) + dy * (k11 * dy);
        double e = exp(-diff2 / 4.0);
        if (HAS_W) e = e * w[i] * w[i + k];
        s += e;
    }
    const double r = block_sum(s, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// =============================================================================================================
extern "C" {

int gd_kde_lag_sums_2d(gd_ctx* ctx, int32_t coli, int32_t colj, const double* kinv3, const int64_t* lags, int32_t nlags,
                       double* out) {
    GD_REQUIRE(ctx && kinv3 && lags && out && nlags > 0 && nlags <= 64, "bad argument");
    GD_REQUIRE(ctx->cols && coli >= 0 && coli < ctx->n + GD_EXTRA_COLS && colj >= 0 && colj < ctx->n + GD_EXTRA_COLS, "bad column");
    for (int i = 0; i < nlags; ++i) GD_REQUIRE(lags[i] > 0 && lags[i] < ctx->N, "lag out of range");
    int nblk = (8 * ctx->cu_count + nlags - 1) / nlags;
    if (nblk < 8) nblk = 8;
    if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
    const int64_t o_l = ((int64_t)nlags * nblk * 8 + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, o_l + nlags * 8);
    if (!base) return GD_ERR_NOMEM;
    double* part = (double*)base;
    int64_t* d_lags = (int64_t*)(base + o_l);
    GD_TRY(gd_h2d(ctx, d_lags, lags, (size_t)nlags * 8));
    const double* x = ctx->cols + (int64_t)coli * ctx->ld;
    const double* y = ctx->cols + (int64_t)colj * ctx->ld;
    const dim3 grid(nblk, nlags);
    // kinv3 = {K00, K01 + K10, K11} of inv(cov)/h^2
    if (ctx->w)
        k_kde_lag_2d<true><<<grid, 256, 0, ctx->stream>>>(x, y, ctx->w, ctx->N, kinv3[0], kinv3[1], kinv3[2], d_lags, part);
    else
        k_kde_lag_2d<false><<<grid, 256, 0, ctx->stream>>>(x, y, nullptr, ctx->N, kinv3[0], kinv3[1], kinv3[2], d_lags, part);
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)nlags * nblk);
    GD_TRY(gd_fetch(ctx, h.data(), part, h.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int l = 0; l < nlags; ++l) {
        double sum = 0;
        for (int b = 0; b < nblk; ++b) sum += h[(size_t)l * nblk + b];
        out[l] = sum;
    }
    return GD_OK;
}

int gd_weight_stats(gd_ctx* ctx, int64_t lo, int64_t hi, double thresh, double* out4) {
    GD_REQUIRE(ctx && out4, "null argument");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    if (!ctx->w) {
        const double cnt = (double)(hi - lo);
        out4[0] = cnt, out4[1] = 1.0, out4[2] = cnt, out4[3] = (1.0 > thresh) ? cnt : 0.0;
        return GD_OK;
    }
    const int nblk = NBLK_STREAM;
    double* part = (double*)gd_scratch(ctx, (int64_t)nblk * 4 * 8);
    if (!part) return GD_ERR_NOMEM;
    k_weight_stats<<<nblk, 256, 0, ctx->stream>>>(ctx->w, lo, hi, thresh, part);
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)nblk * 4);
    GD_TRY(gd_fetch(ctx, h.data(), part, h.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    double s = 0, mx = -INFINITY, s2 = 0, cnt = 0;
    for (int i = 0; i < nblk; ++i) {
        s += h[i * 4], s2 += h[i * 4 + 2], cnt += h[i * 4 + 3];
        if (h[i * 4 + 1] > mx) mx = h[i * 4 + 1];
    }
    out4[0] = s, out4[1] = mx, out4[2] = s2, out4[3] = cnt;
    return GD_OK;
}

// shared by gd_col_stats / gd_cov: d_res[c] = {min,max,sumw,mean}; optional variance into d_out (n x 4)
static int col_stats_device(gd_ctx* ctx, const int32_t* d_colidx, int ncols, int64_t lo, int64_t hi, double* d_res,
                            double* d_part, double* d_out) {
    const int nblk = NBLK_STREAM;
    dim3 grid(nblk, ncols);
    {
        constexpr int G = 8;
        const dim3 ggrid(nblk, (ncols + G - 1) / G);
        if (ctx->w)
            k_col_pass1g<true, G><<<ggrid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_colidx, ncols, ctx->w, lo, hi, d_part);
        else
            k_col_pass1g<false, G><<<ggrid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_colidx, ncols, nullptr, lo, hi, d_part);
    }
    GD_KERNEL_CHECK();
    k_col_fin1<<<ncols, 256, 0, ctx->stream>>>(d_part, nblk, d_res);
    GD_KERNEL_CHECK();
    if (d_out) {
        if (ctx->w)
            k_col_pass2<true><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_colidx, ctx->w, lo, hi, d_res, d_part);
        else
            k_col_pass2<false><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_colidx, nullptr, lo, hi, d_res,
                                                              d_part);
        GD_KERNEL_CHECK();
        k_col_fin2<<<ncols, 256, 0, ctx->stream>>>(d_part, nblk, d_res, d_out);
        GD_KERNEL_CHECK();
    }
    return GD_OK;
}

int gd_col_stats(gd_ctx* ctx, int64_t lo, int64_t hi, double* out) {
    GD_REQUIRE(ctx && out, "null argument");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    const int n = (int)ctx->n;
    const int64_t bytes = ((int64_t)n * NBLK_STREAM * 4 + (int64_t)n * 8) * 8;
    double* base = (double*)gd_scratch(ctx, bytes);
    if (!base) return GD_ERR_NOMEM;
    double* d_part = base;
    double* d_res = base + (int64_t)n * NBLK_STREAM * 4;
    double* d_out = d_res + (int64_t)n * 4;
    int rc = col_stats_device(ctx, nullptr, n, lo, hi, d_res, d_part, d_out);
    if (rc) return rc;
    GD_TRY(gd_fetch(ctx, out, d_out, (size_t)n * 4 * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_col_minmax(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t lo, int64_t hi, int32_t cond_col,
                  double cond_below, double* out) {
    GD_REQUIRE(ctx && cols && out && ncols > 0, "bad argument");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    GD_REQUIRE(cond_col < ctx->n + GD_EXTRA_COLS, "condition column out of range");
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    const int nblk = NBLK_STREAM;
    const int64_t idx_bytes = ((int64_t)ncols * 4 + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, idx_bytes + (int64_t)ncols * nblk * 2 * 8);
    if (!base) return GD_ERR_NOMEM;
    int32_t* d_idx = (int32_t*)base;
    double* d_part = (double*)(base + idx_bytes);
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)ncols * 4));
    dim3 grid(nblk, ncols);
    if (cond_col >= 0)
        k_col_minmax<true><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->cols + (int64_t)cond_col * ctx->ld,
                                                          cond_below, lo, hi, d_part);
    else
        k_col_minmax<false><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, 0.0, lo, hi, d_part);
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)ncols * nblk * 2);
    GD_TRY(gd_fetch(ctx, h.data(), d_part, h.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int c = 0; c < ncols; ++c) {
        double mn = INFINITY, mx = -INFINITY;
        for (int b = 0; b < nblk; ++b) {
            mn = fmin(mn, h[((size_t)c * nblk + b) * 2]);
            mx = fmax(mx, h[((size_t)c * nblk + b) * 2 + 1]);
        }
        out[2 * c] = mn, out[2 * c + 1] = mx;
    }
    return GD_OK;
}

int gd_cov(gd_ctx* ctx, const int32_t* cols, int32_t m, int64_t lo, int64_t hi, double* means_out, double* cov_out,
           double* norm_out, double* minmax_out) {
    GD_REQUIRE(ctx && cols && means_out && cov_out && norm_out && m > 0, "bad argument");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    for (int i = 0; i < m; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    const int64_t rows = hi - lo;
    auto take_init = [](int64_t& off, int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    double* d_res = nullptr;
    double* d_cov = nullptr;
    const bool slab = (m <= 208) && !getenv("GDHIP_COV_TILE");
    if (slab) {
        // one pass (round 6): no means pass in front -- a provisional shift, the ones column, k_cov_onepass_fin (above
        // k_cov_slab2).  m = 208 has no room for the extra column and keeps the two passes.
        const bool onepass = m + 1 <= 208 && !getenv("GDHIP_COV_TWOPASS");
        const int me = onepass ? m + 1 : m;
        // ---- slab kernel: pick the wave arrangement with the fewest tile-pair slots >= T
        const int nt16 = (me + 15) / 16, T = nt16 * (nt16 + 1) / 2, mc = nt16 * 16;
        const int KS = mc <= 64 ? 64 : 32;  // narrow matrices: more rows per barrier pair
        struct Cfg { int mcap, nw, nh, p, bpc; };
        static const Cfg cfgs[] = {{64, 8, 8, 3, 4},  {64, 8, 4, 3, 4},   {64, 8, 4, 5, 4},   {112, 8, 2, 4, 2}, {112, 8, 2, 6, 2},
                                   {112, 8, 2, 7, 2}, {208, 16, 2, 6, 1}, {208, 16, 1, 5, 1}, {208, 16, 1, 6, 1}};
        int pick = -1;
        for (int k = 0; k < (int)(sizeof(cfgs) / sizeof(cfgs[0])); ++k)
            if (cfgs[k].mcap >= mc && (cfgs[k].nw / cfgs[k].nh) * cfgs[k].p >= T) {
                pick = k;
                break;
            }
        GD_REQUIRE(pick >= 0, "covariance: no slab configuration");
        const Cfg cf = cfgs[pick];
        const bool hw = ctx->w != nullptr;
        size_t lds = (size_t)(hw ? 2 : 1) * mc * (KS + 2) * 8 + (onepass ? (size_t)mc * 8 : 0);
        if (cf.nh > 1) lds = std::max(lds, (size_t)(cf.nw / cf.nh) * cf.p * 256 * 8);  // the in-block sum over the row-splits
        int bpc = cf.bpc;
        while (bpc > 1 && (size_t)bpc * lds > 150u * 1024u) --bpc;
        int nblk = bpc * ctx->cu_count;
        if (nblk > (rows + 4 * KS - 1) / (4 * KS)) nblk = (int)((rows + 4 * KS - 1) / (4 * KS));
        if (nblk < 1) nblk = 1;
        int64_t rows_per_chunk = (rows + nblk - 1) / nblk;
        rows_per_chunk = (rows_per_chunk + KS - 1) / KS * KS;
        nblk = (int)((rows + rows_per_chunk - 1) / rows_per_chunk);
        int64_t off = 0;
        const int64_t o_part1 = take_init(off, (int64_t)m * NBLK_STREAM * 4 * 8), o_res = take_init(off, (int64_t)m * 4 * 8),
                      o_idx = take_init(off, (int64_t)m * 4), o_cpart = take_init(off, (int64_t)nblk * T * 256 * 8),
                      o_cov = take_init(off, (int64_t)m * m * 8), o_mm = take_init(off, (int64_t)nblk * m * 2 * 8),
                      o_S = take_init(off, (int64_t)mc * mc * 8);
        char* base = (char*)gd_scratch(ctx, off);
        if (!base) return GD_ERR_NOMEM;
        double* d_part1 = (double*)(base + o_part1);
        d_res = (double*)(base + o_res);
        int32_t* d_idx = (int32_t*)(base + o_idx);
        double* d_cpart = (double*)(base + o_cpart);
        d_cov = (double*)(base + o_cov);
        double* d_mm = (double*)(base + o_mm);
        double* d_S = (double*)(base + o_S);
        GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)m * 4));
        if (onepass) {
            k_col_shift<<<m, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, lo, hi, d_res);
            GD_KERNEL_CHECK();
        } else {
            int rc = col_stats_device(ctx, d_idx, m, lo, hi, d_res, d_part1, nullptr);
            if (rc) return rc;
        }
#define GD_COV2(HW, MCAP, NW, NH, PP)                                                                                     \
    do {                                                                                                                  \
        if (onepass) {                                                                                                    \
            auto kern = k_cov_slab2<HW, MCAP, NW, NH, PP, ((MCAP) <= 64 ? 64 : 32), ((MCAP) <= 64 ? 2 : 1), true>;        \
            GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
            kern<<<nblk, (NW) * 64, lds, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, m, d_res, ctx->w, lo, hi, rows_per_chunk, \
                                                        d_cpart, d_mm);                                                   \
        } else {                                                                                                          \
            auto kern = k_cov_slab2<HW, MCAP, NW, NH, PP, ((MCAP) <= 64 ? 64 : 32), ((MCAP) <= 64 ? 2 : 1), false>;       \
            GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
            kern<<<nblk, (NW) * 64, lds, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, m, d_res, ctx->w, lo, hi, rows_per_chunk, \
                                                        d_cpart, nullptr);                                                \
        }                                                                                                                 \
    } while (0)
#define GD_COV2_HW(MCAP, NW, NH, PP) \
    do {                              \
        if (hw)                       \
            GD_COV2(true, MCAP, NW, NH, PP);  \
        else                          \
            GD_COV2(false, MCAP, NW, NH, PP); \
    } while (0)
        switch (pick) {
            case 0: GD_COV2_HW(64, 8, 8, 3); break;
            case 1: GD_COV2_HW(64, 8, 4, 3); break;
            case 2: GD_COV2_HW(64, 8, 4, 5); break;
            case 3: GD_COV2_HW(112, 8, 2, 4); break;
            case 4: GD_COV2_HW(112, 8, 2, 6); break;
            case 5: GD_COV2_HW(112, 8, 2, 7); break;
            case 6: GD_COV2_HW(208, 16, 2, 6); break;
            case 7: GD_COV2_HW(208, 16, 1, 5); break;
            default: GD_COV2_HW(208, 16, 1, 6); break;
        }
#undef GD_COV2_HW
#undef GD_COV2
        GD_KERNEL_CHECK();
        if (onepass) {
            k_cov_slab_sum<<<T, 1024, 0, ctx->stream>>>(d_cpart, nblk, nt16, d_S);
            GD_KERNEL_CHECK();
            k_cov_onepass_fin<<<m, 256, 0, ctx->stream>>>(d_S, mc, m, d_mm, nblk, d_res, d_cov);
        } else {
            k_cov_slab_fin<<<T, 1024, 0, ctx->stream>>>(d_cpart, nblk, m, d_res, d_cov);
        }
        GD_KERNEL_CHECK();
    } else {
    const int nt = (m + CT - 1) / CT;
    std::vector<int2> tiles;
    for (int a = 0; a < nt; ++a)
        for (int b = a; b < nt; ++b) tiles.push_back(make_int2(a, b));
    const int ntp = (int)tiles.size();
    int nchunks = (2 * ctx->cu_count + ntp - 1) / ntp;
    if (nchunks > (rows + CRB - 1) / CRB) nchunks = (int)((rows + CRB - 1) / CRB);
    if (nchunks < 1) nchunks = 1;
    int64_t rows_per_chunk = (rows + nchunks - 1) / nchunks;
    rows_per_chunk = (rows_per_chunk + CRB - 1) / CRB * CRB;
    nchunks = (int)((rows + rows_per_chunk - 1) / rows_per_chunk);
    // scratch layout
    int64_t off = 0;
    auto take = [&](int64_t bytes) { return take_init(off, bytes); };
    const int64_t o_part1 = take((int64_t)m * NBLK_STREAM * 4 * 8), o_res = take((int64_t)m * 4 * 8),
                  o_idx = take((int64_t)m * 4), o_tiles = take((int64_t)ntp * 8),
                  o_cpart = take((int64_t)ntp * nchunks * CT * CT * 8), o_cov = take((int64_t)m * m * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    double* d_part1 = (double*)(base + o_part1);
    d_res = (double*)(base + o_res);
    int32_t* d_idx = (int32_t*)(base + o_idx);
    int2* d_tiles = (int2*)(base + o_tiles);
    double* d_cpart = (double*)(base + o_cpart);
    d_cov = (double*)(base + o_cov);
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)m * 4));
    GD_TRY(gd_h2d(ctx, d_tiles, tiles.data(), (size_t)ntp * 8));
    int rc = col_stats_device(ctx, d_idx, m, lo, hi, d_res, d_part1, nullptr);
    if (rc) return rc;
    dim3 grid(nchunks, ntp);
    if (ctx->w)
        k_cov_tile<true><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, m, d_res, ctx->w, lo, hi,
                                                         rows_per_chunk, d_tiles, d_cpart);
    else
        k_cov_tile<false><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, m, d_res, nullptr, lo, hi,
                                                          rows_per_chunk, d_tiles, d_cpart);
    GD_KERNEL_CHECK();
    k_cov_fin<<<dim3((CT * CT + 255) / 256, ntp), 256, 0, ctx->stream>>>(d_cpart, nchunks, d_tiles, m, d_res, d_cov);
    GD_KERNEL_CHECK();
    }
    std::vector<double> hres((size_t)m * 4);
    GD_TRY(gd_fetch(ctx, hres.data(), d_res, hres.size() * 8));
    GD_TRY(gd_fetch(ctx, cov_out, d_cov, (size_t)m * m * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int i = 0; i < m; ++i) means_out[i] = hres[(size_t)i * 4 + 3];
    if (minmax_out)
        for (int i = 0; i < m; ++i) minmax_out[2 * i] = hres[(size_t)i * 4], minmax_out[2 * i + 1] = hres[(size_t)i * 4 + 1];
    *norm_out = hres[2];
    return GD_OK;
}

// the linear-bucket path of gd_quantiles_mm; returns 1 when a bucket list overflowed (the caller takes the radix path)
// Bucket columns of the resident sample set (ctx.hpp BucketCols): allocated on first use when the device has room for them
// beside what a large call needs (they are an optimisation: without them the later passes read the fp64 samples).
static unsigned short* bucket_columns(gd_ctx* ctx) {
    if (!ctx->bq || getenv("GDHIP_NO_BUCKET_COLS")) return nullptr;
    BucketCols& B = *ctx->bq;
    std::lock_guard<std::mutex> g(B.mu);
    if (B.buf) return B.buf;
    if (B.tried) return nullptr;
    B.tried = true;
    const size_t bytes = (size_t)ctx->n * ctx->ld * 2;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 4 * bytes + ((size_t)8 << 30)) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    B.buf = (unsigned short*)p;
    B.n = ctx->n, B.ld = ctx->ld;
    B.mn.assign((size_t)ctx->n, 0.0), B.scale.assign((size_t)ctx->n, 0.0), B.nb.assign((size_t)ctx->n, 0);
    return B.buf;
}

static int quantiles_linear(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t lo, int64_t hi, const double* targets,
                            int32_t k, const double* minmax, double* out, int* overflowed, const double* probe_means,
                            double* probe_out) {
    const bool hw = ctx->w != nullptr;
    const int nb = hw ? QLIN_NB_W : QLIN_NB_U;
    std::vector<QState> hst((size_t)ncols);
    std::vector<QLin> hql((size_t)ncols);
    memset(hst.data(), 0, hst.size() * sizeof(QState));
    memset(hql.data(), 0, hql.size() * sizeof(QLin));
    for (int c = 0; c < ncols; ++c) {
        hst[c].k = k;
        hst[c].nuniq = 1;
        for (int t = 0; t < k; ++t) hst[c].target[t] = targets[(size_t)c * k + t];
        hql[c].mn = minmax[2 * c];
        hql[c].scale = (double)nb / (minmax[2 * c + 1] - minmax[2 * c]);
        hql[c].nb = nb;
    }
    // blocks per column: the chip filled about twice over, but every block keeps at least 256K rows to spread its 128-KB
    // table's zero-fill and flush over (few columns -- one rank's share of the parameters -- would otherwise get 64
    // blocks of 150K rows each)
    int nblk = (2 * ctx->cu_count + ncols - 1) / ncols;
    if (nblk > 32) nblk = 32;
    if ((int64_t)nblk * 262144 > hi - lo) nblk = (int)((hi - lo + 262143) / 262144);
    if (nblk < 1) nblk = 1;
    {
        // one block per CU (the table is 128 KB): ncols * nblk blocks run in ceil(blocks / CUs) rounds, so a count just
        // above a multiple of the CU number wastes most of a round (50 columns x 11 blocks = 2.15 rounds took the time of
        // 3; x 10 = 1.95 rounds: 1.40 -> 1.05 ms).  Among the counts down to half of the first choice take the one that
        // fills its rounds best.
        int best = nblk;
        double best_eff = 0;
        for (int q = nblk; q >= (nblk + 1) / 2 && q >= 1; --q) {
            const double rounds = (double)ncols * q / ctx->cu_count;
            const double eff = rounds / ceil(rounds);
            if (eff > best_eff + 0.02) best_eff = eff, best = q;
        }
        nblk = best;
    }
    int nblk2 = (8 * ctx->cu_count + ncols - 1) / ncols;
    if (nblk2 < 8) nblk2 = 8;
    if (nblk2 > 2 * ctx->cu_count) nblk2 = 2 * ctx->cu_count;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_st = take((int64_t)ncols * sizeof(QState)), o_ql = take((int64_t)ncols * sizeof(QLin)),
                  o_idx = take((int64_t)ncols * 4), o_out = take((int64_t)ncols * k * 8),
                  o_cnt = take((int64_t)ncols * QK_MAX * 4 + 256), o_lk = take((int64_t)ncols * QK_MAX * QCAP * 8),
                  o_lw = take((int64_t)ncols * QK_MAX * QCAP * 8), o_part = take((int64_t)ncols * nblk * nb * (hw ? 8 : 4)),
                  o_tot = take((int64_t)ncols * nb * 8), o_pmean = take((int64_t)ncols * 8),
                  o_ppart = take((int64_t)ncols * nblk * QP_LAGS * 8), o_pout = take((int64_t)ncols * QP_LAGS * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    // whole columns of the sample set: the fused pass (tiles; bucket columns for the later passes; the lag probe on request)
    bool whole = lo == 0 && hi == ctx->N && !getenv("GDHIP_QLIN_UNFUSED");
    for (int c = 0; whole && c < ncols; ++c) whole = cols[c] < ctx->n;
    unsigned short* bq = whole ? bucket_columns(ctx) : nullptr;
    const bool probe = whole && probe_means && probe_out;
    double* d_pmean = (double*)(base + o_pmean);
    double* d_ppart = (double*)(base + o_ppart);
    double* d_pout = (double*)(base + o_pout);
    if (probe) GD_TRY(gd_h2d(ctx, d_pmean, probe_means, (size_t)ncols * 8));
    if (bq) {  // the columns' buckets are being rewritten: not valid until this call has seen its kernels complete
        std::lock_guard<std::mutex> g(ctx->bq->mu);
        for (int c = 0; c < ncols; ++c) ctx->bq->nb[cols[c]] = 0;
    }
    QState* d_st = (QState*)(base + o_st);
    QLin* d_ql = (QLin*)(base + o_ql);
    int32_t* d_idx = (int32_t*)(base + o_idx);
    double* d_out = (double*)(base + o_out);
    int* d_cnt = (int*)(base + o_cnt);
    unsigned long long* d_lk = (unsigned long long*)(base + o_lk);
    double* d_lw = (double*)(base + o_lw);
    void* d_part = base + o_part;
    double* d_tot = (double*)(base + o_tot);
    GD_TRY(gd_h2d(ctx, d_st, hst.data(), hst.size() * sizeof(QState)));
    GD_TRY(gd_h2d(ctx, d_ql, hql.data(), hql.size() * sizeof(QLin)));
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)ncols * 4));
    GD_HIP(hipMemsetAsync(d_cnt, 0, (size_t)ncols * QK_MAX * 4 + 256, ctx->stream));
    const size_t lds = (size_t)nb * (hw ? 8 : 4);
    const size_t lds_tiles = lds + (probe ? (size_t)16 * (128 + QP_LAGS) * 8 : 0);
    // 2^k with 2^k * (sum of all weights) < 2^62 (gd_quantiles_mm checked that the sum is known and positive)
    const double wscale = hw ? ldexp(1.0, 61 - ilogb(ctx->w_sum)) : 1.0;
#define GD_QLIN(HW)                                                                                                                  \
    do {                                                                                                                             \
        if (whole) {                                                                                                                 \
            if (probe) {                                                                                                             \
                GD_HIP(hipFuncSetAttribute((const void*)k_qlin_count_tiles<HW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tiles)); \
                k_qlin_count_tiles<HW, true><<<dim3(nblk, ncols), 1024, lds_tiles, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_ql, \
                                                                                                 d_part, wscale, bq, d_pmean, d_ppart);  \
            } else {                                                                                                                 \
                GD_HIP(hipFuncSetAttribute((const void*)k_qlin_count_tiles<HW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tiles)); \
                k_qlin_count_tiles<HW, false><<<dim3(nblk, ncols), 1024, lds_tiles, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_ql, \
                                                                                                  d_part, wscale, bq, nullptr, nullptr); \
            }                                                                                                                        \
        } else {                                                                                                                     \
            GD_HIP(hipFuncSetAttribute((const void*)k_qlin_count<HW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
            k_qlin_count<HW><<<dim3(nblk, ncols), 1024, lds, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, lo, hi, d_ql, d_part, wscale); \
        }                                                                                                                            \
        GD_KERNEL_CHECK();                                                                                                           \
        if (probe) {                                                                                                                 \
            k_sum_partials_batched<<<dim3(QP_LAGS, ncols), 256, 0, ctx->stream>>>(d_ppart, nblk, QP_LAGS, d_pout);                   \
            GD_KERNEL_CHECK();                                                                                                       \
        }                                                                                                                            \
        k_qlin_reduce<HW><<<dim3(nb / 256, ncols), 256, 0, ctx->stream>>>(d_part, nblk, d_tot, 1.0 / wscale);                        \
        GD_KERNEL_CHECK();                                                                                                           \
        k_qlin_scan<HW><<<ncols, 1024, 0, ctx->stream>>>(d_st, d_ql, d_tot);                                                         \
        GD_KERNEL_CHECK();                                                                                                           \
        if (bq)                                                                                                                      \
            k_qlin_collect_bq<HW><<<dim3(nblk2, ncols), 512, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_st, d_ql, bq, d_lk, \
                                                                               d_lw, d_cnt);                                         \
        else                                                                                                                         \
            k_qlin_collect<HW><<<dim3(nblk2, ncols), 512, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, lo, hi, d_st, d_ql, d_lk, d_lw,  \
                                                                            d_cnt);                                                  \
    } while (0)
    if (hw)
        GD_QLIN(true);
    else
        GD_QLIN(false);
#undef GD_QLIN
    GD_KERNEL_CHECK();
    k_qsel_finish<<<dim3(ncols, QK_MAX), 1024, 0, ctx->stream>>>(d_st, d_lk, d_lw, d_cnt, k, 8, d_out, d_cnt + (int64_t)ncols * QK_MAX,
                                                                 hw || getenv("GDHIP_QSEL_WALK") ? 0 : 1);
    GD_KERNEL_CHECK();
    int overflow = 0;
    GD_TRY(gd_fetch(ctx, &overflow, d_cnt + (int64_t)ncols * QK_MAX, 4));
    GD_TRY(gd_fetch(ctx, out, d_out, (size_t)ncols * k * 8));
    if (probe) GD_TRY(gd_fetch(ctx, probe_out, d_pout, (size_t)ncols * QP_LAGS * 8));
    GD_TRY(gd_stream_sync(ctx));
    if (bq) {
        std::lock_guard<std::mutex> g(ctx->bq->mu);
        for (int c = 0; c < ncols; ++c)
            ctx->bq->mn[cols[c]] = hql[c].mn, ctx->bq->scale[cols[c]] = hql[c].scale, ctx->bq->nb[cols[c]] = nb;
    }
    *overflowed = overflow;
    return GD_OK;
}

int gd_quantiles_mm(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t lo, int64_t hi, const double* targets, int32_t k,
                    const double* minmax, double* out) {
    return gd_quantiles_mm_probe(ctx, cols, ncols, lo, hi, targets, k, minmax, out, nullptr, nullptr, nullptr);
}

int gd_quantiles_mm_probe(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t lo, int64_t hi, const double* targets, int32_t k,
                          const double* minmax, double* out, const double* probe_means, double* probe_out, int32_t* probe_done) {
    GD_REQUIRE(ctx && cols && targets && out && ncols > 0, "bad argument");
    if (probe_done) *probe_done = 0;
    GD_REQUIRE(k > 0 && k <= QK_MAX, "at most 16 quantiles per call");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    bool linear = minmax != nullptr && getenv("GDHIP_QSEL_RADIX") == nullptr;
    // the expected length of a live bucket's list is rows / buckets times the peak-to-mean density ratio (about 4 for a
    // Gaussian over its sampled range); beyond these row counts the lists would overflow QCAP
    if (linear && (hi - lo) > (ctx->w ? 12000000 : 25000000)) linear = false;
    // weighted bucket sums are fixed-point integers scaled from the weights' total, which is known for the uploaded sample
    // weights (any kind: multiplicities or real); auxiliary weights (gd_select_weights) take the radix path
    if (linear && ctx->w && !(ctx->w_sum > 1e-200 && ctx->w_sum < 1e200)) linear = false;
    for (int c = 0; linear && c < ncols; ++c) {
        const double a = minmax[2 * c], b = minmax[2 * c + 1];
        if (!(b > a) || !std::isfinite(a) || !std::isfinite(b) || !std::isfinite((double)QLIN_NB_U / (b - a))) linear = false;
    }
    if (linear) {
        for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
        int overflowed = 0;
        const bool want_probe = probe_means && probe_out && probe_done && lo == 0 && hi == ctx->N && !getenv("GDHIP_QLIN_UNFUSED");
        const int rc = quantiles_linear(ctx, cols, ncols, lo, hi, targets, k, minmax, out, &overflowed, want_probe ? probe_means : nullptr,
                                        want_probe ? probe_out : nullptr);
        if (rc == GD_OK && want_probe) {
            bool whole = true;  // (the fused pass ran: every column is one of the sample set's own)
            for (int c = 0; c < ncols; ++c) whole = whole && cols[c] < ctx->n;
            *probe_done = whole ? 1 : 0;
        }
        if (rc || !overflowed) return rc;  // heavily tied or very peaked data: the radix path below redoes the call
    }
    return gd_quantiles(ctx, cols, ncols, lo, hi, targets, k, out);
}

int gd_quantiles(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t lo, int64_t hi, const double* targets,
                 int32_t k, double* out) {
    GD_REQUIRE(ctx && cols && targets && out && ncols > 0, "bad argument");
    GD_REQUIRE(k > 0 && k <= QK_MAX, "at most 16 quantiles per call");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    std::vector<QState> hst((size_t)ncols);
    memset(hst.data(), 0, hst.size() * sizeof(QState));
    for (int c = 0; c < ncols; ++c) {
        hst[c].k = k;
        hst[c].nuniq = 1;
        for (int t = 0; t < k; ++t) hst[c].target[t] = targets[(size_t)c * k + t];
    }
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_st = take((int64_t)ncols * sizeof(QState)), o_h = take((int64_t)ncols * QK_MAX * 256 * 8),
                  o_idx = take((int64_t)ncols * 4), o_out = take((int64_t)ncols * k * 8),
                  o_cnt = take((int64_t)ncols * QK_MAX * 4 + 256), o_lk = take((int64_t)ncols * QK_MAX * QCAP * 8),
                  o_lw = take((int64_t)ncols * QK_MAX * QCAP * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    QState* d_st = (QState*)(base + o_st);
    double* d_h = (double*)(base + o_h);
    int32_t* d_idx = (int32_t*)(base + o_idx);
    double* d_out = (double*)(base + o_out);
    int* d_cnt = (int*)(base + o_cnt);
    unsigned long long* d_lk = (unsigned long long*)(base + o_lk);
    double* d_lw = (double*)(base + o_lw);
    GD_TRY(gd_h2d(ctx, d_st, hst.data(), hst.size() * sizeof(QState)));
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)ncols * 4));
    GD_HIP(hipMemsetAsync(d_h, 0, (size_t)ncols * QK_MAX * 256 * 8, ctx->stream));
    // few blocks per column: every block flushes its private LDS histograms with global atomics on the same
    // per-column bins, so the flush cost (and its contention) grows with the block count
    int nblk = (8 * ctx->cu_count + ncols - 1) / ncols;
    if (nblk < 8) nblk = 8;
    if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
    auto radix_pass = [&](int pass) -> int {
        dim3 grid(nblk, ncols);
        if (ctx->w)
            k_qsel_pass<true><<<grid, 512, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, lo, hi, pass, d_st, d_h);
        else
            k_qsel_pass<false><<<grid, 512, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, lo, hi, pass, d_st,
                                                               d_h);
        GD_KERNEL_CHECK();
        k_qsel_scan<<<ncols, 256, 0, ctx->stream>>>(d_st, d_h, ncols, pass);
        GD_KERNEL_CHECK();
        return GD_OK;
    };
    // radix passes until the live buckets are short, then collect + sort them (k_qsel_collect / k_qsel_finish)
    const int P = (hi - lo) <= 15000000 ? 3 : 4;
    int rc;
    for (int pass = 0; pass < P; ++pass)
        if ((rc = radix_pass(pass))) return rc;
    GD_HIP(hipMemsetAsync(d_cnt, 0, (size_t)ncols * QK_MAX * 4 + 256, ctx->stream));
    {
        dim3 grid(nblk, ncols);
        if (ctx->w)
            k_qsel_collect<true><<<grid, 512, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, lo, hi, P, d_st, d_lk, d_lw,
                                                                  d_cnt);
        else
            k_qsel_collect<false><<<grid, 512, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, lo, hi, P, d_st, d_lk,
                                                                   d_lw, d_cnt);
        GD_KERNEL_CHECK();
    }
    k_qsel_finish<<<dim3(ncols, QK_MAX), 1024, 0, ctx->stream>>>(d_st, d_lk, d_lw, d_cnt, k, P, d_out, d_cnt + (int64_t)ncols * QK_MAX,
                                                                 ctx->w || getenv("GDHIP_QSEL_WALK") ? 0 : 1);
    GD_KERNEL_CHECK();
    int overflow = 0;
    GD_TRY(gd_fetch(ctx, &overflow, d_cnt + (int64_t)ncols * QK_MAX, 4));
    GD_TRY(gd_fetch(ctx, out, d_out, (size_t)ncols * k * 8));
    GD_TRY(gd_stream_sync(ctx));
    if (!overflow) return GD_OK;
    for (int pass = P; pass < 8; ++pass)  // heavily tied data: finish with the plain radix passes
        if ((rc = radix_pass(pass))) return rc;
    k_qsel_out<<<(ncols * k + 255) / 256, 256, 0, ctx->stream>>>(d_st, ncols, k, d_out);
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, out, d_out, (size_t)ncols * k * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_autocov_lags_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* means, int64_t k0,
                          int32_t nlags, double* out) {
    GD_REQUIRE(ctx, "null context");
    return gd_autocov_lags_range_batch(ctx, cols, ncols, means, 0, ctx->N, k0, nlags, out);
}

int gd_autocov_lags_range_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* means, int64_t row_lo,
                                int64_t row_hi, int64_t k0, int32_t nlags, double* out) {
    GD_REQUIRE(ctx && cols && means && out && nlags > 0 && ncols > 0, "bad argument");
    GD_REQUIRE(ctx->cols && k0 >= 0, "bad lag / no samples");
    GD_REQUIRE(row_lo >= 0 && row_hi <= ctx->N && row_lo < row_hi, "bad row range");
    const int64_t NR = row_hi - row_lo;
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    int nblk = (8 * ctx->cu_count + ncols - 1) / ncols;
    if (nblk < 16) nblk = 16;
    if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_part = take((int64_t)ncols * nblk * AL * 8), o_out = take((int64_t)ncols * AL * 8),
                  o_idx = take((int64_t)ncols * 4), o_mean = take((int64_t)ncols * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    double* part = (double*)(base + o_part);
    double* d_out = (double*)(base + o_out);
    int32_t* d_idx = (int32_t*)(base + o_idx);
    double* d_mean = (double*)(base + o_mean);
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)ncols * 4));
    GD_TRY(gd_h2d(ctx, d_mean, means, (size_t)ncols * 8));
    std::vector<double> h((size_t)ncols * AL);
    for (int32_t done = 0; done < nlags;) {
        const dim3 grid(nblk, ncols);
        const int NL = (nlags - done <= 8) ? 8 : AL;  // short first probe: uncorrelated chains stop at lag 1
        if (NL == 8) {
            if (ctx->w)
                k_autocov<true, 8><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, row_lo, NR, d_mean, k0 + done, part);
            else
                k_autocov<false, 8><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, row_lo, NR, d_mean, k0 + done, part);
        } else {
            if (ctx->w)
                k_autocov<true, AL><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, row_lo, NR, d_mean, k0 + done, part);
            else
                k_autocov<false, AL><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, row_lo, NR, d_mean, k0 + done, part);
        }
        GD_KERNEL_CHECK();
        k_sum_partials_batched<<<dim3(NL, ncols), 256, 0, ctx->stream>>>(part, nblk, NL, d_out);
        GD_KERNEL_CHECK();
        GD_TRY(gd_fetch(ctx, h.data(), d_out, (size_t)ncols * NL * 8));
        GD_TRY(gd_stream_sync(ctx));
        const int take_n = (nlags - done < NL) ? nlags - done : NL;
        for (int c = 0; c < ncols; ++c)
            for (int l = 0; l < take_n; ++l) out[(size_t)c * nlags + done + l] = h[(size_t)c * NL + l];
        done += take_n;
    }
    return GD_OK;
}

int gd_autocov_lags(gd_ctx* ctx, int32_t col, double mean, int64_t k0, int32_t nlags, double* out) {
    return gd_autocov_lags_batch(ctx, &col, 1, &mean, k0, nlags, out);
}

int gd_kde_lag_sums_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* inv4s2, const int64_t* lags,
                          int32_t nlags, double* out) {
    GD_REQUIRE(ctx && cols && inv4s2 && lags && out && nlags > 0 && nlags <= 64 && ncols > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    for (int i = 0; i < nlags; ++i) GD_REQUIRE(lags[i] > 0 && lags[i] < ctx->N, "lag out of range");
    const bool multi = nlags <= KDE_LAG_MAX && getenv("GDHIP_KDE_LAG_SINGLE") == nullptr;
    int per_cu = 8;  // blocks of 256 threads per CU (GDHIP_KDE_LAG_BLOCKS_PER_CU: tuning knob)
    if (const char* e = getenv("GDHIP_KDE_LAG_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    int nblk = multi ? (per_cu * ctx->cu_count + ncols - 1) / ncols : (8 * ctx->cu_count + ncols * nlags - 1) / (ncols * nlags);
    if (nblk < 8) nblk = 8;
    if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_part = take((int64_t)ncols * (multi ? KDE_LAG_MAX : nlags) * nblk * 8), o_lags = take((int64_t)nlags * 8),
                  o_idx = take((int64_t)ncols * 4), o_c = take((int64_t)ncols * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    double* part = (double*)(base + o_part);
    int64_t* d_lags = (int64_t*)(base + o_lags);
    int32_t* d_idx = (int32_t*)(base + o_idx);
    double* d_c = (double*)(base + o_c);
    // the N_eff lag set (five lags from N/2, one or two short ones): every sample read once
    int order[KDE_LAG_MAX];
    int nf = 0, nn = 0;
    bool folded = false;
    if (multi && ctx->N >= 4096 && getenv("GDHIP_KDE_LAG_UNFOLDED") == nullptr) {
        int64_t K0 = ctx->N, nmax = 0;
        for (int i = 0; i < nlags; ++i)
            if (4 * lags[i] >= ctx->N) {
                order[nf++] = i;
                K0 = lags[i] < K0 ? lags[i] : K0;
            }
        for (int i = 0; i < nlags; ++i)
            if (4 * lags[i] < ctx->N) {
                order[nf + nn++] = i;
                nmax = lags[i] > nmax ? lags[i] : nmax;
            }
        folded = nf == 5 && (nn == 1 || nn == 2) && 2 * K0 <= ctx->N && nmax < 256;
    }
    if (folded) {
        int64_t sorted[KDE_LAG_MAX];
        for (int i = 0; i < nlags; ++i) sorted[i] = lags[order[i]];
        GD_TRY(gd_h2d(ctx, d_lags, sorted, (size_t)nlags * 8));
    } else {
        for (int i = 0; i < nlags && i < KDE_LAG_MAX; ++i) order[i] = i;
        GD_TRY(gd_h2d(ctx, d_lags, lags, (size_t)nlags * 8));
    }
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)ncols * 4));
    GD_TRY(gd_h2d(ctx, d_c, inv4s2, (size_t)ncols * 8));
    if (multi) {
        const dim3 grid(nblk, ncols);
        if (folded) {
#define KDE_FOLDED(W, NNV)                                                                                           \
    k_kde_lag_folded<W, 5, NNV><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_c, d_lags, part)
            if (ctx->w) {
                if (nn == 1) KDE_FOLDED(true, 1); else KDE_FOLDED(true, 2);
            } else {
                if (nn == 1) KDE_FOLDED(false, 1); else KDE_FOLDED(false, 2);
            }
#undef KDE_FOLDED
        } else
#define KDE_LAUNCH(NLV)                                                                                                 \
    case NLV:                                                                                                           \
        if (ctx->w)                                                                                                     \
            k_kde_lag_multi<true, NLV><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_c,   \
                                                                      d_lags, part);                                    \
        else                                                                                                            \
            k_kde_lag_multi<false, NLV><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, ctx->N, d_c, \
                                                                       d_lags, part);                                   \
        break;
        switch (nlags) {
            KDE_LAUNCH(1) KDE_LAUNCH(2) KDE_LAUNCH(3) KDE_LAUNCH(4) KDE_LAUNCH(5) KDE_LAUNCH(6) KDE_LAUNCH(7) KDE_LAUNCH(8)
        }
#undef KDE_LAUNCH
        GD_KERNEL_CHECK();
        std::vector<double> h((size_t)ncols * nblk * KDE_LAG_MAX);
        GD_TRY(gd_fetch(ctx, h.data(), part, h.size() * 8));
        GD_TRY(gd_stream_sync(ctx));
        for (int c = 0; c < ncols; ++c)
            for (int l = 0; l < nlags; ++l) {
                double sum = 0;
                for (int b = 0; b < nblk; ++b) sum += h[((size_t)c * nblk + b) * KDE_LAG_MAX + l];
                out[(size_t)c * nlags + order[l]] = sum;
            }
        return GD_OK;
    }
    const dim3 grid(nblk, nlags, ncols);
    if (ctx->w)
        k_kde_lag<true><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_c, d_lags, part);
    else
        k_kde_lag<false><<<grid, 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, ctx->N, d_c, d_lags, part);
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)ncols * nlags * nblk);
    GD_TRY(gd_fetch(ctx, h.data(), part, h.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (size_t q = 0; q < (size_t)ncols * nlags; ++q) {
        double sum = 0;
        for (int b = 0; b < nblk; ++b) sum += h[q * nblk + b];
        out[q] = sum;
    }
    return GD_OK;
}

int gd_kde_lag_sums(gd_ctx* ctx, int32_t col, double inv4s2, const int64_t* lags, int32_t nlags, double* out) {
    return gd_kde_lag_sums_batch(ctx, &col, 1, &inv4s2, lags, nlags, out);
}

}  // extern "C"
