// Weighted binning: fused bin-index computation + LDS-privatised histograms (1D and stripe-tiled 2D).
//
// Index arithmetic is the reference's, in IEEE fp64 with true division and no FMA contraction (this
// file is compiled with -ffp-contract=off): ix = (int)((x - binmin) / width + 0.5)  (mcsamples.py:1497)
// and the truncating form (int)((x - range_min) / dx) of kde_bandwidth.py:86-87.
//
// 2D layout: an F x F fp64 histogram (512 KB at F=256) does not fit the 160 KB LDS, so a workgroup owns a
// stripe of R rows (R*F bins <= 128 KB of LDS), scans its chunk of samples and accumulates only the
// samples whose y index falls in the stripe.  The stripes of one (pair, chunk) unit are given block ids
// that are congruent mod 8, i.e. land on the same XCD, so that the repeated reads of the same sample
// range are served by that XCD's L2.  Unit-weight sample sets use u32 LDS counters (twice the rows per
// stripe, native ds_add_u32); weighted ones use fp64 LDS atomics (ds_add_f64).
#include "ctx.hpp"

#include <array>
#include <map>

#define LDS_HIST_BYTES (128 * 1024)

// The quotient (x - binmin) / width must be the correctly rounded IEEE quotient (bin indices are bit-exact against
// numpy), but the divisor is the same for every sample of a column: with y = RN(1 / width) taken once per thread by a
// true division, q <- RN(n y), then twice  r = n - q width (one exact FMA), q <- RN(q + r y)  yields RN(n / width) for
// every n (Markstein's correction step: exact unless the divisor's significand is all ones, which takes the true
// division below).  5 fp64 operations per sample instead of the ~30 of the division sequence (v_rcp_f64 at quarter
// rate, three Newton steps, scale / fixup); checked against n / d on 4e8 random and near-half-integer cases on the host.
struct BinDiv {
    double binmin, width, rcp;
};
__device__ __forceinline__ BinDiv make_bindiv(double binmin, double width) {
    BinDiv d;
    d.binmin = binmin, d.width = width;
    const unsigned long long mant = (unsigned long long)__double_as_longlong(width) & 0xFFFFFFFFFFFFFull;
    d.rcp = (mant == 0xFFFFFFFFFFFFFull) ? 0.0 : 1.0 / width;  // 0: use the true division
    return d;
}
__device__ __forceinline__ double bin_quotient(double x, const BinDiv& d) {
    const double n = x - d.binmin;
    double q = n * d.rcp;
    double r = fma(-q, d.width, n);
    q = fma(r, d.rcp, q);
    r = fma(-q, d.width, n);
    q = fma(r, d.rcp, q);
    // infinities / NaN (r is NaN then) and the all-ones divisor: the division itself
    if (!(fabs(q) < 1e300) || d.rcp == 0.0) q = n / d.width;
    return q;
}
__device__ __forceinline__ int bin_round(double x, const BinDiv& d) {
    return (int)(bin_quotient(x, d) + 0.5);
}
__device__ __forceinline__ int bin_trunc(double x, const BinDiv& d) {
    return (int)bin_quotient(x, d);
}

// ---- 1D -----------------------------------------------------------------------------------------------
// grid (nblk, ncols); 256 threads = 4 waves, one private histogram copy per wave.
template <bool HAS_W>
__global__ void __launch_bounds__(256) k_hist1d(const double* __restrict__ cols, int64_t ld,
                                                const int32_t* __restrict__ colidx, const double* __restrict__ w,
                                                int64_t N, const double* __restrict__ binmin,
                                                const double* __restrict__ width, int F, double* __restrict__ part) {
    extern __shared__ double sh[];  // 4 * F
    const int c = blockIdx.y;
    const double* x = cols + (int64_t)colidx[c] * ld;
    const BinDiv bd = make_bindiv(binmin[c], width[c]);
    for (int i = threadIdx.x; i < 4 * F; i += 256) sh[i] = 0;
    __syncthreads();
    double* mine = sh + (threadIdx.x >> 6) * F;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    const int64_t Ne = N & ~(int64_t)1;
    for (int64_t i = 2 * gtid; i < Ne; i += 2 * gsz) {
        const double2 xv = *reinterpret_cast<const double2*>(x + i);
        double2 wv = make_double2(1.0, 1.0);
        if (HAS_W) wv = *reinterpret_cast<const double2*>(w + i);
        const int i0 = bin_round(xv.x, bd), i1 = bin_round(xv.y, bd);
        if ((unsigned)i0 < (unsigned)F) atomicAdd(&mine[i0], wv.x);
        if ((unsigned)i1 < (unsigned)F) atomicAdd(&mine[i1], wv.y);
    }
    if (gtid == 0 && Ne < N) {
        const int i0 = bin_round(x[Ne], bd);
        if ((unsigned)i0 < (unsigned)F) atomicAdd(&mine[i0], HAS_W ? w[Ne] : 1.0);
    }
    __syncthreads();
    double* p = part + ((int64_t)c * gridDim.x + blockIdx.x) * F;
    for (int i = threadIdx.x; i < F; i += 256) p[i] = (sh[i] + sh[F + i]) + (sh[2 * F + i] + sh[3 * F + i]);
}

// out[c][f] = sum_b part[c][b][f]
__global__ void k_hist1d_reduce(const double* __restrict__ part, int nblk, int F, double* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double* p = part + (int64_t)blockIdx.y * nblk * F + f;
    double s = 0;
    for (int b = 0; b < nblk; ++b) s += p[(int64_t)b * F];
    out[(int64_t)blockIdx.y * F + f] = s;
}

template <bool ROUND, typename T>
__global__ void k_bin_indices(const double* __restrict__ x, int64_t N, double binmin, double width, int F,
                              T* __restrict__ idx, unsigned long long* __restrict__ n_bad) {
    unsigned long long bad = 0;
    const BinDiv bd = make_bindiv(binmin, width);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = ROUND ? bin_round(x[i], bd) : bin_trunc(x[i], bd);
        idx[i] = (T)v;
        bad += ((unsigned)v >= (unsigned)F);
    }
    if (n_bad && bad) atomicAdd(n_bad, bad);
}

// u16 bin indices, 8 samples (4 x double2 in, one 16-byte store out) per thread-iteration.
__global__ void __launch_bounds__(256) k_prebin(const double* __restrict__ x, int64_t N, double binmin, double width,
                                                int F, unsigned short* __restrict__ idx) {
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    const int64_t N8 = N & ~(int64_t)7;
    const BinDiv bd = make_bindiv(binmin, width);
    for (int64_t i = 8 * gtid; i < N8; i += 8 * gsz) {
        double2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const double2*>(x + i + 2 * q);
        unsigned short o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = bin_round(v[q].x, bd), b = bin_round(v[q].y, bd);
            o[2 * q] = (unsigned)a < (unsigned)F ? (unsigned short)a : (unsigned short)0xFFFF;
            o[2 * q + 1] = (unsigned)b < (unsigned)F ? (unsigned short)b : (unsigned short)0xFFFF;
        }
        uint4 pk;
        pk.x = o[0] | ((unsigned)o[1] << 16);
        pk.y = o[2] | ((unsigned)o[3] << 16);
        pk.z = o[4] | ((unsigned)o[5] << 16);
        pk.w = o[6] | ((unsigned)o[7] << 16);
        *reinterpret_cast<uint4*>(idx + i) = pk;
    }
    if (gtid == 0)
        for (int64_t i = N8; i < N; ++i) {
            const int a = bin_round(x[i], bd);
            idx[i] = (unsigned)a < (unsigned)F ? (unsigned short)a : (unsigned short)0xFFFF;
        }
}

struct PrebinCol {
    const double* x;
    unsigned short* idx;
    double binmin, width;
};

// all requested index columns in one launch; grid (blocks, ncols)
__global__ void __launch_bounds__(256) k_prebin_batch(const PrebinCol* __restrict__ colsv, int64_t N, int F) {
    const PrebinCol C = colsv[blockIdx.y];
    const BinDiv bd = make_bindiv(C.binmin, C.width);
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    const int64_t N8 = N & ~(int64_t)7;
    for (int64_t i = 8 * gtid; i < N8; i += 8 * gsz) {
        double2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gload_d2(C.x + i + 2 * q);
        unsigned int o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = bin_round(v[q].x, bd), b = bin_round(v[q].y, bd);
            o[2 * q] = (unsigned)a < (unsigned)F ? (unsigned)a : 0xFFFFu;
            o[2 * q + 1] = (unsigned)b < (unsigned)F ? (unsigned)b : 0xFFFFu;
        }
        uint4 pk;
        pk.x = o[0] | (o[1] << 16);
        pk.y = o[2] | (o[3] << 16);
        pk.z = o[4] | (o[5] << 16);
        pk.w = o[6] | (o[7] << 16);
        *reinterpret_cast<uint4*>(C.idx + i) = pk;
    }
    if (gtid == 0)
        for (int64_t i = N8; i < N; ++i) {
            const int a = bin_round(C.x[i], bd);
            C.idx[i] = (unsigned)a < (unsigned)F ? (unsigned short)a : (unsigned short)0xFFFF;
        }
}

// ---- 2D -----------------------------------------------------------------------------------------------
struct Hist2DPair {   // per-pair parameters, device array
    const double* x;  // column for the x index (or x_i of the shear)
    const double* y;  // column for the y index (or x_j of the shear)
    const unsigned short* ix;  // prebinned variants
    const unsigned short* iy;
    double bx, wx, by, wy;  // bin origin / width for x and y
    double r0, r1;          // shear: y value = r0*x + r1*y
};

template <typename BinT>
__device__ __forceinline__ void lds_add(BinT* p, double w);
template <>
__device__ __forceinline__ void lds_add<double>(double* p, double w) {
    atomicAdd(p, w);
}
template <>
__device__ __forceinline__ void lds_add<unsigned int>(unsigned int* p, double w) {
    atomicAdd(p, (unsigned int)w);  // unit weights (w = 1) or integral weights (exact)
}

// decode the 1-D grid: stripes of one unit share an XCD (block id mod 8)
__device__ __forceinline__ void decode_block(int nstripes, int nchunks, int& pair, int& chunk, int& stripe) {
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    stripe = q % nstripes;
    const int unit = (q / nstripes) * 8 + xcd;
    pair = unit / nchunks;
    chunk = unit % nchunks;
}
// Tile-major order: the blocks of ONE row tile (all pairs) are neighbours in the grid, so the pairs that share a column
// read the same megabyte of it at the same time and all but the first find it in the memory-side cache; pair >= B marks
// the padding blocks.
__device__ __forceinline__ void decode_block_tiles(int B, int nstripes, int nchunks, int& pair, int& chunk, int& stripe) {
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    stripe = q % nstripes;
    const int unit = (q / nstripes) * 8 + xcd;
    chunk = unit / B;
    pair = chunk < nchunks ? unit % B : B;
}

template <typename BinT>
__device__ __forceinline__ void flush_stripe(const BinT* sh, int row0, int R, int F, double* __restrict__ hist,
                                             bool exclusive) {
    const int nb = R * F;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        const int r = row0 + i / F;
        if (r >= F) break;
        const double v = (double)sh[i];
        double* dst = hist + (int64_t)r * F + (i % F);
        if (exclusive)
            *dst = v;
        else if (v != 0)
            unsafeAtomicAdd(dst, v);
    }
}

// MODE 0: rounded indices from two fp64 columns; MODE 1: sheared (truncating) indices; MODE 2: prebinned u16.
template <int MODE, bool HAS_W, typename BinT>
__global__ void __launch_bounds__(1024) k_hist2d(const Hist2DPair* __restrict__ pairs, int B,
                                                 const double* __restrict__ w, int64_t N, int F, int R, int nstripes,
                                                 int nchunks, double* __restrict__ hist_all) {
    extern __shared__ double sh_raw[];
    BinT* sh = reinterpret_cast<BinT*>(sh_raw);
    int pair, chunk, stripe;
    decode_block(nstripes, nchunks, pair, chunk, stripe);
    if (pair >= B) return;
    const Hist2DPair P = pairs[pair];
    const BinDiv bdx = make_bindiv(P.bx, P.wx), bdy = make_bindiv(P.by, P.wy);
    const int row0 = stripe * R;
    for (int i = threadIdx.x; i < R * F; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    // rows of this chunk, aligned to 8 samples
    int64_t per = (N + nchunks - 1) / nchunks;
    per = (per + 7) & ~(int64_t)7;
    const int64_t lo = (int64_t)chunk * per;
    int64_t hi = lo + per;
    if (hi > N) hi = N;
    if (MODE == 2) {
        const int64_t hi8 = lo + ((hi - lo) & ~(int64_t)7);
        for (int64_t i = lo + 8 * (int64_t)threadIdx.x; i < hi8; i += 8 * (int64_t)blockDim.x) {
            const uint4 ax = gload_u4(P.ix + i);
            const uint4 ay = gload_u4(P.iy + i);
            double2 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wv[q] = HAS_W ? *reinterpret_cast<const double2*>(w + i + 2 * q) : make_double2(1.0, 1.0);
            const unsigned xs[4] = {ax.x, ax.y, ax.z, ax.w}, ys[4] = {ay.x, ay.y, ay.z, ay.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned cx = (xs[q] >> (16 * h)) & 0xFFFFu, cy = (ys[q] >> (16 * h)) & 0xFFFFu;
                    const unsigned r = cy - (unsigned)row0;
                    if (r < (unsigned)R && cx < (unsigned)F && cy < (unsigned)F)
                        lds_add<BinT>(&sh[r * F + cx], h ? wv[q].y : wv[q].x);
                }
            }
        }
        if (threadIdx.x == 0)
            for (int64_t i = hi8; i < hi; ++i) {
                const unsigned cx = P.ix[i], cy = P.iy[i];
                const unsigned r = cy - (unsigned)row0;
                if (r < (unsigned)R && cx < (unsigned)F && cy < (unsigned)F)
                    lds_add<BinT>(&sh[r * F + cx], HAS_W ? w[i] : 1.0);
            }
    } else {
        const int64_t hi2 = lo + ((hi - lo) & ~(int64_t)1);
        for (int64_t i = lo + 2 * (int64_t)threadIdx.x; i < hi2; i += 2 * (int64_t)blockDim.x) {
            const double2 xv = gload_d2(P.x + i);
            const double2 yv = gload_d2(P.y + i);
            double2 wv = make_double2(1.0, 1.0);
            if (HAS_W) wv = *reinterpret_cast<const double2*>(w + i);
            int cx0, cy0, cx1, cy1;
            if (MODE == 0) {
                cx0 = bin_round(xv.x, bdx), cy0 = bin_round(yv.x, bdy);
                cx1 = bin_round(xv.y, bdx), cy1 = bin_round(yv.y, bdy);
            } else {
                cx0 = bin_trunc(xv.x, bdx), cx1 = bin_trunc(xv.y, bdx);
                const double p0 = P.r0 * xv.x + P.r1 * yv.x, p1 = P.r0 * xv.y + P.r1 * yv.y;
                cy0 = bin_trunc(p0, bdy), cy1 = bin_trunc(p1, bdy);
            }
            unsigned r = (unsigned)cy0 - (unsigned)row0;
            if (r < (unsigned)R && (unsigned)cx0 < (unsigned)F && (unsigned)cy0 < (unsigned)F)
                lds_add<BinT>(&sh[r * F + cx0], wv.x);
            r = (unsigned)cy1 - (unsigned)row0;
            if (r < (unsigned)R && (unsigned)cx1 < (unsigned)F && (unsigned)cy1 < (unsigned)F)
                lds_add<BinT>(&sh[r * F + cx1], wv.y);
        }
        if (threadIdx.x == 0 && hi2 < hi) {
            const double xv = P.x[hi2], yv = P.y[hi2];
            int cx, cy;
            if (MODE == 0) {
                cx = bin_round(xv, bdx), cy = bin_round(yv, bdy);
            } else {
                cx = bin_trunc(xv, bdx);
                cy = bin_trunc(P.r0 * xv + P.r1 * yv, bdy);
            }
            const unsigned r = (unsigned)cy - (unsigned)row0;
            if (r < (unsigned)R && (unsigned)cx < (unsigned)F && (unsigned)cy < (unsigned)F)
                lds_add<BinT>(&sh[r * F + cx], HAS_W ? w[hi2] : 1.0);
        }
    }
    __syncthreads();
    flush_stripe<BinT>(sh, row0, R, F, hist_all + (int64_t)pair * F * F, nchunks == 1);
}


// Unit-weight, pre-binned variant with 16-bit LDS counters packed two per word: a 256 x 256 grid fits one
// 128 KB stripe, so every sample is visited once instead of once per stripe.  A counter can wrap if a bin receives
// more than 65535 samples; a wrap always lowers the sum of all counters, so comparing that sum with the number of
// samples the block accepted detects it exactly, and the host redoes flagged pairs with the 32-bit kernel.
// HAS_W8: integer multiplicities <= 255 ride along as one byte per sample (exact: the counters add integers; the
// wrap test compares against the accepted WEIGHT instead of the accepted count).
template <bool HAS_W8>
__global__ void __launch_bounds__(1024) k_hist2d_u16(const Hist2DPair* __restrict__ pairs, int B, int64_t N, int F,
                                                     int R, int nstripes, const unsigned char* __restrict__ w8,
                                                     double* __restrict__ hist_all, int* __restrict__ overflow) {
    extern __shared__ double sh_raw[];
    unsigned int* sh = reinterpret_cast<unsigned int*>(sh_raw);
    __shared__ double red[16];
    int pair, chunk, stripe;
    decode_block(nstripes, 1, pair, chunk, stripe);
    if (pair >= B) return;
    const Hist2DPair P = pairs[pair];
    const BinDiv bdx = make_bindiv(P.bx, P.wx), bdy = make_bindiv(P.by, P.wy);
    const int row0 = stripe * R;
    const int nwords = (R * F + 1) / 2;
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    unsigned int nacc = 0;
    auto visit = [&](unsigned cx, unsigned cy, unsigned wt) {
        const unsigned r = cy - (unsigned)row0;
        if (r < (unsigned)R && cx < (unsigned)F && cy < (unsigned)F && (!HAS_W8 || wt != 0u)) {
            const unsigned a = r * (unsigned)F + cx;
            atomicAdd(&sh[a >> 1], wt << ((a & 1u) * 16u));
            nacc += wt;
        }
    };
    // 8 samples: two u16 per word of the index columns, one byte per sample of weights (wlo = samples 0-3, whi = 4-7)
    auto visit8 = [&](const uint4& ax, const uint4& ay, unsigned wlo, unsigned whi) {
        const unsigned xs[4] = {ax.x, ax.y, ax.z, ax.w}, ys[4] = {ay.x, ay.y, ay.z, ay.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned ww = (q < 2 ? wlo : whi) >> ((q & 1) * 16);
            visit(xs[q] & 0xFFFFu, ys[q] & 0xFFFFu, HAS_W8 ? (ww & 0xFFu) : 1u);
            visit(xs[q] >> 16, ys[q] >> 16, HAS_W8 ? ((ww >> 8) & 0xFFu) : 1u);
        }
    };
    const int64_t N16 = N & ~(int64_t)15;
    // two 16-byte loads per index column in flight before any sample is consumed
    for (int64_t i = 16 * (int64_t)threadIdx.x; i < N16; i += 16 * (int64_t)blockDim.x) {
        const uint4 ax0 = gload_u4(P.ix + i), ay0 = gload_u4(P.iy + i);
        const uint4 ax1 = gload_u4(P.ix + i + 8), ay1 = gload_u4(P.iy + i + 8);
        uint4 wv = make_uint4(0, 0, 0, 0);
        if (HAS_W8) wv = *reinterpret_cast<const uint4*>(w8 + i);
        visit8(ax0, ay0, wv.x, wv.y);
        visit8(ax1, ay1, wv.z, wv.w);
    }
    if (threadIdx.x == 0)
        for (int64_t i = N16; i < N; ++i) visit(P.ix[i], P.iy[i], HAS_W8 ? (unsigned)w8[i] : 1u);
    __syncthreads();
    double* hist = hist_all + (int64_t)pair * F * F;
    unsigned int total = 0;
    for (int i = threadIdx.x; i < R * F; i += blockDim.x) {
        const int r = row0 + i / F;
        if (r >= F) break;
        const unsigned int v = (sh[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
        total += v;
        hist[(int64_t)r * F + (i % F)] = (double)v;
    }
    const double t = block_sum((double)total, red), a = block_sum((double)nacc, red);
    if (threadIdx.x == 0 && t != a) atomicOr(&overflow[pair], 1);
}

// ---- F = 256, unit weights, byte indices: the batched triangle kernel ------------------------------------------------
// One block per pair; the 256 x 256 grid is 65536 16-bit counters packed two per LDS word (128 KB).  Bin a = (y << 8) | x
// lives in word (a & 0x7fff), half (a >> 15): neighbouring x bins fall in neighbouring banks, and the byte-interleave
// v_perm_b32 builds two bin addresses per instruction from four x bytes and four y bytes, so a sample costs about half
// the VALU instructions of the u16 kernel (which is issue-bound as much as LDS-bound: profiles/).  Every sample is
// inside the grid by construction of the bin range (the pre-bin kernel verifies it), so the accepted count is N and
// a counter wrap -- which always changes the sum of all counters -- is detected by comparing that sum with N.
struct Hist2DPair8 {
    const unsigned char* ix;
    const unsigned char* iy;
};

// Memory side: the index bytes come in by global_load (the per-pair table holds generic pointers, and a flat load also
// counts on lgkmcnt -- waiting for one drains all 16 LDS atomics of the previous iteration), DEPTH iterations of loads in
// flight per lane (register ring: a wave always has adds to issue while its next bytes travel); the packed counters sit
// at LDS address 0 and the increment is 1 + 0xffff * (top bit of y): 4.5 VALU operations per sample.
// (profiles/r03_kernels_ab.json: 3.43 ms for the flat-load, no-ring form of round 2 -> 2.88 ms at DEPTH 3; 1, 2, 4: 3.19,
// 3.00, 2.89 ms.)
// nchunks > 1 (round 5; a rank's share of a triangle, or any call with fewer pairs than twice the CUs): the rows of a pair
// are cut into nchunks ranges, one block each, the packed counters go to `part` as they are and k_p8_reduce adds the
// chunks -- 125 pairs on 256 CUs used to leave half the chip idle for the whole launch, 294 pairs took two rounds where
// 1.15 would do.  Counter wraps are detected per chunk as before; the sum over chunks is formed in 32 bits.
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_hist2d_u8_pf(const Hist2DPair8* __restrict__ pairs, int B, int64_t N,
                                                       double* __restrict__ hist_all, int* __restrict__ overflow, int nchunks = 1,
                                                       unsigned int* __restrict__ part = nullptr) {
    extern __shared__ double sh_raw[];  // 32768 words of counters, then 16 doubles for the block reduction
    unsigned int* sh = reinterpret_cast<unsigned int*>(sh_raw);
    double* red = sh_raw + 16384;
    const int units = B * nchunks, per_xcd = (units + 7) / 8;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (unit >= units || (int)(blockIdx.x >> 3) >= per_xcd) return;
    const int pair = unit / nchunks, chunk = unit - pair * nchunks;
    const Hist2DPair8 P = pairs[pair];
    const int64_t nvec_all = N >> 4;  // 16-byte vectors per column
    int64_t per = (nvec_all + nchunks - 1) / nchunks;
    per = (per + 1023) & ~(int64_t)1023;
    const int64_t v0 = min(nvec_all, (int64_t)chunk * per), v1 = (nchunks == 1) ? nvec_all : min(nvec_all, v0 + per);
    const bool last = chunk == nchunks - 1;
    const uint4* gx = reinterpret_cast<const uint4*>(P.ix) + v0;
    const uint4* gy = reinterpret_cast<const uint4*>(P.iy) + v0;
    for (int i = threadIdx.x; i < 32768; i += 1024) sh[i] = 0;
    __syncthreads();
    auto visit2 = [&](unsigned v) {  // v = (y1 x1 y0 x0): two samples
        atomicAdd(&sh[v & 0x7fffu], __builtin_amdgcn_ubfe(v, 15, 1) * 0xffffu + 1u);
        atomicAdd(&sh[__builtin_amdgcn_ubfe(v, 16, 15)], (v >> 31) * 0xffffu + 1u);
    };
    auto visit16 = [&](const uint4& ax, const uint4& ay) {
        const unsigned xs[4] = {ax.x, ax.y, ax.z, ax.w}, ys[4] = {ay.x, ay.y, ay.z, ay.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            visit2(__builtin_amdgcn_perm(ys[q], xs[q], 0x05010400u));
            visit2(__builtin_amdgcn_perm(ys[q], xs[q], 0x07030602u));
        }
    };
    const int64_t nvec = v1 - v0;  // this block's vectors
    const int64_t K = (nvec >> 10) / DEPTH * DEPTH;  // rounds in which every lane has a vector, a multiple of the ring depth
    if (K > 0) {
        uint4 rx[DEPTH], ry[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            rx[d] = gload_u4(gx + d * 1024 + threadIdx.x), ry[d] = gload_u4(gy + d * 1024 + threadIdx.x);
            __builtin_amdgcn_sched_barrier(0);  // in slot order, or the loop head must wait for the youngest load
        }
        for (int64_t k = 0; k < K; k += DEPTH) {
            // the rounds these slots serve next; past the end: a harmless re-read of the last ring
            const int64_t kn = (k + DEPTH < K ? k + DEPTH : K - DEPTH) * 1024 + threadIdx.x;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                visit16(rx[d], ry[d]);  // consumed first: the reload lands in the same registers, no copies at the back edge
                rx[d] = gload_u4(gx + kn + d * 1024), ry[d] = gload_u4(gy + kn + d * 1024);
                // keep the slots apart: merged by the scheduler, all DEPTH loads would be waited for at the loop head
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (int64_t u = K * 1024 + threadIdx.x; u < nvec; u += 1024) visit16(gload_u4(gx + u), gload_u4(gy + u));
    if (threadIdx.x == 0 && last)
        for (int64_t i = nvec_all << 4; i < N; ++i) {
            const unsigned a = ((unsigned)P.iy[i] << 8) | (unsigned)P.ix[i];
            atomicAdd(&sh[a & 0x7fffu], (a >> 15) * 0xffffu + 1u);
        }
    __syncthreads();
    if (nchunks > 1) {  // the packed counters as they are; k_p8_reduce adds the chunks
        unsigned int* out = part + (int64_t)unit * 32768;
        unsigned int total = 0;
        for (int i = 4 * threadIdx.x; i < 32768; i += 4096) {
            const uint4 c = *reinterpret_cast<const uint4*>(&sh[i]);
            total += (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu) + (c.y >> 16) + (c.z & 0xffffu) + (c.z >> 16) + (c.w & 0xffffu) + (c.w >> 16);
            *reinterpret_cast<uint4*>(&out[i]) = c;
        }
        const double t = block_sum((double)total, red);
        const double mine = (double)((v1 - v0) << 4) + (last ? (double)(N - (nvec_all << 4)) : 0.0);
        if (threadIdx.x == 0 && t != mine) atomicOr(&overflow[pair], 1);
        return;
    }
    double* hist = hist_all + (int64_t)pair * 65536;
    unsigned int total = 0;
    for (int i = 2 * threadIdx.x; i < 32768; i += 2048) {  // two words per lane: 16-byte stores (the tail is store-issue bound)
        const uint2 c = *reinterpret_cast<const uint2*>(&sh[i]);
        total += (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu) + (c.y >> 16);
        *reinterpret_cast<double2*>(&hist[i]) = make_double2((double)(c.x & 0xffffu), (double)(c.y & 0xffffu));
        *reinterpret_cast<double2*>(&hist[i + 32768]) = make_double2((double)(c.x >> 16), (double)(c.y >> 16));
    }
    const double t = block_sum((double)total, red);
    if (threadIdx.x == 0 && t != (double)N) atomicOr(&overflow[pair], 1);
}

// hist[pair][a] = sum over chunks of the low (a < 32768) / high halves of word a & 0x7fff.  grid (32, B) x 256: four words per lane
__global__ void __launch_bounds__(256) k_p8_reduce(const unsigned int* __restrict__ part, int nchunks, double* __restrict__ hist_all) {
    const int pair = blockIdx.y;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    const unsigned int* p = part + (int64_t)pair * nchunks * 32768 + i;
    unsigned int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    for (int c = 0; c < nchunks; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + (int64_t)c * 32768);
        lo[0] += v.x & 0xffffu, hi[0] += v.x >> 16, lo[1] += v.y & 0xffffu, hi[1] += v.y >> 16;
        lo[2] += v.z & 0xffffu, hi[2] += v.z >> 16, lo[3] += v.w & 0xffffu, hi[3] += v.w >> 16;
    }
    double* hist = hist_all + (int64_t)pair * 65536;
    *reinterpret_cast<double2*>(&hist[i]) = make_double2((double)lo[0], (double)lo[1]);
    *reinterpret_cast<double2*>(&hist[i + 2]) = make_double2((double)lo[2], (double)lo[3]);
    *reinterpret_cast<double2*>(&hist[i + 32768]) = make_double2((double)hi[0], (double)hi[1]);
    *reinterpret_cast<double2*>(&hist[i + 32768 + 2]) = make_double2((double)hi[2], (double)hi[3]);
}

// chunks per pair of the byte-index launch: none for a whole triangle (the partial tables would cost more than the round
// quantisation), else what fills the rounds (pick_chunks)
static int pick_chunks(const gd_ctx* ctx, int64_t units, int64_t rows);
static int u8_chunks(const gd_ctx* ctx, int B, int64_t N) {
    if (B >= 2 * ctx->cu_count) return 1;
    return pick_chunks(ctx, B, N);
}

struct PrebinCol8 {
    const double* x;
    unsigned char* idx;
    double binmin, width;
};

// byte bin indices of several columns in one launch (grid (blocks, ncols)); bad[c] counts samples outside [0, F)
__global__ void __launch_bounds__(256) k_prebin8_batch(const PrebinCol8* __restrict__ colsv, int64_t N, int F,
                                                       unsigned long long* __restrict__ bad) {
    const PrebinCol8 C = colsv[blockIdx.y];
    const BinDiv bd = make_bindiv(C.binmin, C.width);
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    const int64_t N8 = N & ~(int64_t)7;
    unsigned nbad = 0;
    for (int64_t i = 8 * gtid; i < N8; i += 8 * gsz) {
        double2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gload_d2(C.x + i + 2 * q);
        unsigned o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = bin_round(v[q].x, bd), b = bin_round(v[q].y, bd);
            nbad += ((unsigned)a >= (unsigned)F) + ((unsigned)b >= (unsigned)F);
            o[2 * q] = (unsigned)a & 0xffu;
            o[2 * q + 1] = (unsigned)b & 0xffu;
        }
        uint2 pk;
        pk.x = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        pk.y = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
        *reinterpret_cast<uint2*>(C.idx + i) = pk;
    }
    if (gtid == 0)
        for (int64_t i = N8; i < N; ++i) {
            const int a = bin_round(C.x[i], bd);
            nbad += ((unsigned)a >= (unsigned)F);
            C.idx[i] = (unsigned char)a;
        }
    if (nbad) atomicAdd(&bad[blockIdx.y], (unsigned long long)nbad);
}

// ---- index columns from the BUCKET columns of the quantile select (round 6; ctx.hpp BucketCols) ----------------------------
// The select's counting pass left bucket = clamp((int)((x - mn) scale), 0, nb - 1) of every sample (2 bytes).  The bin index
// ix = (int)((x - binmin) / width + 0.5) is a non-decreasing function of x (a correctly rounded quotient by a positive
// divisor, an addition, a truncation), so a bucket whose whole x interval -- widened on both sides by far more than the
// rounding of the bucket arithmetic -- has ONE index at both ends has that index for every sample in it: a table look-up
// replaces the 8-byte read and the division.  Buckets that straddle a bin edge (F - 1 edges among nb buckets: about 1 %
// of the samples), the two end buckets (they also hold whatever was clamped) and anything that maps outside [0, F) take
// the exact fp64 route on the sample itself, so every index is bit-equal to k_prebin8_batch's / k_prebin_batch's.
struct PrebinColB {
    const double* x;
    const unsigned short* bq;
    void* idx;
    double binmin, width;
    double mn, inv_scale, slack;  // bucket b holds x in [mn + b inv_scale, mn + (b + 1) inv_scale), up to rounding
    int nb, pad;
};

// lut[c][b] = the bin index of every sample of bucket b, or 0xFFFF = decide on the sample.  grid (nbmax / 256, ncols)
__global__ void __launch_bounds__(256) k_bucket_lut(const PrebinColB* __restrict__ colsv, int F, int nbmax,
                                                    unsigned short* __restrict__ lut) {
    const PrebinColB C = colsv[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= C.nb) return;
    unsigned short v = 0xFFFF;
    if (b > 0 && b < C.nb - 1) {
        const BinDiv bd = make_bindiv(C.binmin, C.width);
        const double x_lo = (C.mn + ((double)b - 1e-3) * C.inv_scale) - C.slack;
        const double x_hi = (C.mn + ((double)b + 1.0 + 1e-3) * C.inv_scale) + C.slack;
        const int a = bin_round(x_lo, bd), c = bin_round(x_hi, bd);
        if (a == c && (unsigned)a < (unsigned)F) v = (unsigned short)a;
    }
    lut[(int64_t)blockIdx.y * nbmax + b] = v;
}

// OUT8: byte indices + out-of-range count (k_prebin8_batch's contract); else u16 indices with the 0xFFFF sentinel
// (k_prebin_batch's).  grid (blocks, ncols), 512 threads, the column's table in LDS; block b owns the rows
// [b R, (b + 1) R) (R a multiple of 8).  A sample whose bucket does not decide its bin is NOT resolved here -- a dependent
// 8-byte load inside the loop would expose one memory latency per wave and iteration (nearly every 512-sample iteration of a
// wave holds one) -- its row goes to the block's list and k_prebin_fix settles the lists afterwards, all loads independent.
template <bool OUT8>
__global__ void __launch_bounds__(512) k_prebin_bq(const PrebinColB* __restrict__ colsv, int64_t N, int64_t R, int cap, int nbmax,
                                                   const unsigned short* __restrict__ lut, unsigned int* __restrict__ lists,
                                                   int* __restrict__ counts) {
    extern __shared__ unsigned short slut[];
    __shared__ int nlist;
    const PrebinColB C = colsv[blockIdx.y];
    {
        const uint4* src = reinterpret_cast<const uint4*>(lut + (int64_t)blockIdx.y * nbmax);
        uint4* dst = reinterpret_cast<uint4*>(slut);
        for (int i = threadIdx.x; i < C.nb / 8; i += 512) dst[i] = src[i];
    }
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    const int mask = C.nb - 1;
    unsigned int* mylist = lists + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * cap;
    const int64_t g_lo = (int64_t)blockIdx.x * R / 8;
    int64_t g_hi = g_lo + R / 8;
    if (g_hi > N / 8) g_hi = N / 8;  // whole groups of 8 only; the last N % 8 rows are k_prebin_fix's
    auto group = [&](const uint4& v, int64_t g) {
        const unsigned wd[4] = {v.x, v.y, v.z, v.w};
        unsigned o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned r = slut[(wd[e >> 1] >> (16 * (e & 1))) & mask];
            if (r == 0xFFFFu) {
                const int pos = atomicAdd(&nlist, 1);
                if (pos < cap) mylist[pos] = (unsigned)(8 * g + e);
                r = 0;
            }
            o[e] = r;
        }
        if (OUT8) {
            uint2 pk;
            pk.x = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            pk.y = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
            *reinterpret_cast<uint2*>((unsigned char*)C.idx + 8 * g) = pk;
        } else {
            uint4 pk;
            pk.x = o[0] | (o[1] << 16), pk.y = o[2] | (o[3] << 16), pk.z = o[4] | (o[5] << 16), pk.w = o[6] | (o[7] << 16);
            *reinterpret_cast<uint4*>((unsigned short*)C.idx + 8 * g) = pk;
        }
    };
    constexpr int U = 4;
    int64_t i = g_lo + threadIdx.x;
    for (; i + (U - 1) * 512 < g_hi; i += U * 512) {
        uint4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = gload_u4(C.bq + 8 * (i + q * 512));
#pragma unroll
        for (int q = 0; q < U; ++q) group(v[q], i + q * 512);
    }
    for (; i < g_hi; i += 512) group(gload_u4(C.bq + 8 * i), i);
    __syncthreads();
    if (threadIdx.x == 0) counts[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = nlist;
}

// settles what k_prebin_bq left open, with the exact fp64 expression on the sample itself: the rows of each block's list (a
// list that overflowed -- heavily tied data sitting on a bin edge -- means the block's whole row range is redone), and the
// last N % 8 rows.  Same grid as k_prebin_bq.
template <bool OUT8>
__global__ void __launch_bounds__(256) k_prebin_fix(const PrebinColB* __restrict__ colsv, int64_t N, int64_t R, int cap, int F,
                                                    const unsigned int* __restrict__ lists, const int* __restrict__ counts,
                                                    unsigned long long* __restrict__ bad) {
    const PrebinColB C = colsv[blockIdx.y];
    const BinDiv bd = make_bindiv(C.binmin, C.width);
    unsigned nbad = 0;
    auto settle = [&](int64_t row) {
        const int a = bin_round(C.x[row], bd);
        if (OUT8) {
            nbad += ((unsigned)a >= (unsigned)F);
            ((unsigned char*)C.idx)[row] = (unsigned char)a;
        } else {
            ((unsigned short*)C.idx)[row] = (unsigned)a < (unsigned)F ? (unsigned short)a : (unsigned short)0xFFFF;
        }
    };
    const int cnt = counts[(int64_t)blockIdx.y * gridDim.x + blockIdx.x];
    if (cnt <= cap) {
        const unsigned int* mylist = lists + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * cap;
        // four list entries per thread and iteration, their rows and samples requested before the first is settled (every
        // entry is a dependent chain list -> sample -> store of its own: one at a time exposed two round trips per entry)
        int k = threadIdx.x;
        for (; k + 3 * 256 < cnt; k += 4 * 256) {
            unsigned int r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = mylist[k + q * 256];
            double xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = C.x[r[q]];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = bin_round(xv[q], bd);
                if (OUT8) {
                    nbad += ((unsigned)a >= (unsigned)F);
                    ((unsigned char*)C.idx)[r[q]] = (unsigned char)a;
                } else {
                    ((unsigned short*)C.idx)[r[q]] = (unsigned)a < (unsigned)F ? (unsigned short)a : (unsigned short)0xFFFF;
                }
            }
        }
        for (; k < cnt; k += 256) settle((int64_t)mylist[k]);
    } else {
        const int64_t lo = (int64_t)blockIdx.x * R;
        int64_t hi = lo + R;
        if (hi > N / 8 * 8) hi = N / 8 * 8;
        for (int64_t r = lo + threadIdx.x; r < hi; r += 256) settle(r);
    }
    if (blockIdx.x == gridDim.x - 1)
        for (int64_t r = N / 8 * 8 + threadIdx.x; r < N; r += 256) settle(r);
    if (OUT8 && nbad) atomicAdd(&bad[blockIdx.y], (unsigned long long)nbad);
}

// Host side of the bucket route: which of the requested columns have valid bucket columns, their records, and the launches.
// `which[c]` = slot among the bucket-route columns or -1.  Everything is enqueued on ctx->stream; the records and tables
// live in `base` (device scratch the caller reserved: bucket_route_bytes()).
struct BucketRoute {
    std::vector<PrebinColB> recs;
    std::vector<int> slot_of;  // per requested column
    int nbmax = 0;
};
static BucketRoute bucket_route(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width,
                                void* const* d_idx) {
    BucketRoute R;
    R.slot_of.assign((size_t)ncols, -1);
    if (!ctx->bq || getenv("GDHIP_NO_BUCKET_COLS")) return R;
    {
        // the route costs three small launches and a table per column before it saves anything: below a handful of columns
        // (the single index columns of an up-scaled grid class) the fp64 kernel, one launch, is quicker (measured: 12 columns
        // 0.33 against 0.25 ms, 50 columns 0.69 against 0.93 ms)
        const char* e = getenv("GDHIP_BUCKET_ROUTE_MIN");
        const int min_cols = e ? atoi(e) : 24;
        if (ncols < min_cols) return R;
    }
    BucketCols& B = *ctx->bq;
    std::lock_guard<std::mutex> g(B.mu);
    if (!B.buf || B.ld != ctx->ld) return R;
    for (int c = 0; c < ncols; ++c) {
        const int col = cols[c];
        if (col < 0 || col >= B.n || B.nb[col] <= 0) continue;
        PrebinColB r;
        memset(&r, 0, sizeof r);
        r.x = ctx->cols + (int64_t)col * ctx->ld;
        r.bq = B.buf + (int64_t)col * B.ld;
        r.idx = d_idx[c];
        r.binmin = binmin[c], r.width = width[c];
        r.mn = B.mn[col];
        r.inv_scale = 1.0 / B.scale[col];
        const double mx = r.mn + (double)B.nb[col] * r.inv_scale;
        r.slack = 8.0 * 2.220446049250313e-16 * fmax(fabs(r.mn), fabs(mx));
        r.nb = B.nb[col];
        if (!(r.inv_scale > 0) || !std::isfinite(r.inv_scale) || !std::isfinite(r.slack)) continue;
        R.slot_of[c] = (int)R.recs.size();
        R.recs.push_back(r);
        if (r.nb > R.nbmax) R.nbmax = r.nb;
    }
    return R;
}
// launch shape of the bucket route: blocks per column and rows per block (a multiple of 4096: whole iterations of 512
// threads x 8 rows), list capacity per block
struct BucketShape {
    int nblk;
    int64_t R;
    int cap;
};
static BucketShape bucket_shape(const gd_ctx* ctx, int nc) {
    BucketShape S;
    int nblk = (4 * ctx->cu_count + nc - 1) / nc;  // two blocks per CU hold their tables
    int64_t R = (ctx->N + nblk - 1) / nblk;
    if (R < 131072) R = 131072;  // a block's 64-KB table against >= 256 KB of buckets
    R = (R + 4095) / 4096 * 4096;
    nblk = (int)((ctx->N + R - 1) / R);
    if (nblk < 1) nblk = 1;
    S.nblk = nblk, S.R = R, S.cap = (int)(R / 8);
    if (const char* e = getenv("GDHIP_BUCKET_LIST_CAP"))  // test hook: tiny lists, every block takes the redo path
        if (atoi(e) > 0 && atoi(e) < S.cap) S.cap = atoi(e);
    return S;
}
static int64_t bucket_route_bytes(const gd_ctx* ctx, const BucketRoute& R) {
    if (R.recs.empty()) return 0;
    const int nc = (int)R.recs.size();
    const BucketShape S = bucket_shape(ctx, nc);
    return ((int64_t)nc * sizeof(PrebinColB) + 255) / 256 * 256 + ((int64_t)nc * R.nbmax * 2 + 255) / 256 * 256 +
           ((int64_t)nc * 8 + 255) / 256 * 256 + ((int64_t)nc * S.nblk * 4 + 255) / 256 * 256 + (int64_t)nc * S.nblk * S.cap * 4 + 256;
}
// the out-of-range counters (R.recs.size() of them, zeroed here) come back through bad_dev
template <bool OUT8>
static int bucket_route_launch(gd_ctx* ctx, const BucketRoute& R, char* base, int F, unsigned long long** bad_dev) {
    const int nc = (int)R.recs.size();
    const BucketShape S = bucket_shape(ctx, nc);
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_recs = take((int64_t)nc * sizeof(PrebinColB)), o_lut = take((int64_t)nc * R.nbmax * 2), o_bad = take((int64_t)nc * 8),
                  o_cnt = take((int64_t)nc * S.nblk * 4), o_list = take((int64_t)nc * S.nblk * S.cap * 4);
    PrebinColB* d_recs = (PrebinColB*)(base + o_recs);
    unsigned short* d_lut = (unsigned short*)(base + o_lut);
    unsigned long long* d_bad = (unsigned long long*)(base + o_bad);
    int* d_cnt = (int*)(base + o_cnt);
    unsigned int* d_list = (unsigned int*)(base + o_list);
    GD_TRY(gd_h2d(ctx, d_recs, R.recs.data(), (size_t)nc * sizeof(PrebinColB)));
    GD_HIP(hipMemsetAsync(d_bad, 0, (size_t)nc * 8, ctx->stream));
    k_bucket_lut<<<dim3(R.nbmax / 256, nc), 256, 0, ctx->stream>>>(d_recs, F, R.nbmax, d_lut);
    GD_KERNEL_CHECK();
    const size_t lds = (size_t)R.nbmax * 2;
    GD_HIP(hipFuncSetAttribute((const void*)k_prebin_bq<OUT8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_prebin_bq<OUT8><<<dim3(S.nblk, nc), 512, lds, ctx->stream>>>(d_recs, ctx->N, S.R, S.cap, R.nbmax, d_lut, d_list, d_cnt);
    GD_KERNEL_CHECK();
    k_prebin_fix<OUT8><<<dim3(S.nblk, nc), 256, 0, ctx->stream>>>(d_recs, ctx->N, S.R, S.cap, F, d_list, d_cnt, d_bad);
    GD_KERNEL_CHECK();
    *bad_dev = d_bad;
    return GD_OK;
}

// ---- fused fp64 index + binning with packed 16-bit counters (unit weights): one stripe at F <= 256 --------------------
// MODE 0: rounded indices of two columns; MODE 1: the sheared, truncating indices of kde.bin_samples.  A block owns a
// (pair, chunk of rows, stripe of R rows); its packed counters go to scratch and k_p16_reduce adds the chunks, so no
// global atomics and every sample's two divisions are done once per stripe (ONE stripe at F <= 256; the 32-bit kernel
// needs two and therefore reads and divides twice).
#ifndef P16_UNROLL
#define P16_UNROLL 4
#endif
template <int MODE>
__global__ void __launch_bounds__(1024) k_hist2d_f64_p16(const Hist2DPair* __restrict__ pairs, int B, int64_t N, int F, int R,
                                                         int nstripes, int nchunks, int tile_major,
                                                         unsigned int* __restrict__ part, int* __restrict__ overflow) {
    extern __shared__ double sh_raw[];
    unsigned int* sh = reinterpret_cast<unsigned int*>(sh_raw);
    __shared__ double red[16];
    int pair, chunk, stripe;
    if (tile_major)
        decode_block_tiles(B, nstripes, nchunks, pair, chunk, stripe);
    else
        decode_block(nstripes, nchunks, pair, chunk, stripe);
    if (pair >= B) return;
    const Hist2DPair P = pairs[pair];
    const BinDiv bdx = make_bindiv(P.bx, P.wx), bdy = make_bindiv(P.by, P.wy);
    const int row0 = stripe * R;
    const int nwords = (R * F + 1) / 2;
    for (int i = threadIdx.x; i < nwords; i += 1024) sh[i] = 0;
    __syncthreads();
    int64_t per = (N + nchunks - 1) / nchunks;
    per = (per + 7) & ~(int64_t)7;
    const int64_t lo = (int64_t)chunk * per;
    int64_t hi = lo + per;
    if (hi > N) hi = N;
    if (hi < lo) hi = lo;  // (a tile behind the last row)
    unsigned int nacc = 0;
    auto visit = [&](double xv, double yv) {
        int cx, cy;
        if (MODE == 0) {
            cx = bin_round(xv, bdx), cy = bin_round(yv, bdy);
        } else {
            cx = bin_trunc(xv, bdx);
            cy = bin_trunc(P.r0 * xv + P.r1 * yv, bdy);
        }
        const unsigned r = (unsigned)cy - (unsigned)row0;
        if (r < (unsigned)R && (unsigned)cx < (unsigned)F && (unsigned)cy < (unsigned)F) {
            const unsigned a = r * (unsigned)F + (unsigned)cx;
            atomicAdd(&sh[a >> 1], 1u << ((a & 1u) * 16u));
            nacc += 1;
        }
    };
    const int64_t hi2 = lo + ((hi - lo) & ~(int64_t)1);
    // P16_UNROLL pairs of 16-byte loads per lane are requested before the first sample is binned (the index arithmetic of
    // a sample is ~60 fp64 operations: with one pair of loads in flight the lane waits a memory latency per two samples)
    int64_t i = lo + 2 * (int64_t)threadIdx.x;
    for (; i + (int64_t)(P16_UNROLL - 1) * 2 * 1024 < hi2; i += (int64_t)P16_UNROLL * 2 * 1024) {
        double2 xv[P16_UNROLL], yv[P16_UNROLL];
#pragma unroll
        for (int u = 0; u < P16_UNROLL; ++u) {
            xv[u] = gload_d2(P.x + i + (int64_t)u * 2 * 1024);
            yv[u] = gload_d2(P.y + i + (int64_t)u * 2 * 1024);
        }
#pragma unroll
        for (int u = 0; u < P16_UNROLL; ++u) {
            visit(xv[u].x, yv[u].x);
            visit(xv[u].y, yv[u].y);
        }
    }
    for (; i < hi2; i += 2 * 1024) {
        const double2 xv = gload_d2(P.x + i);
        const double2 yv = gload_d2(P.y + i);
        visit(xv.x, yv.x);
        visit(xv.y, yv.y);
    }
    if (threadIdx.x == 0 && hi2 < hi) visit(P.x[hi2], P.y[hi2]);
    __syncthreads();
    unsigned int* dst = part + (((int64_t)pair * nchunks + chunk) * nstripes + stripe) * (int64_t)nwords;
    unsigned int total = 0;
    for (int i = threadIdx.x; i < nwords; i += 1024) {
        const unsigned int v = sh[i];
        total += (v & 0xffffu) + (v >> 16);
        dst[i] = v;
    }
    const double t = block_sum((double)total, red), a = block_sum((double)nacc, red);
    if (threadIdx.x == 0 && t != a) atomicOr(&overflow[pair], 1);
}

// The same packed-counter partials from pre-binned u16 index columns: the few pairs of an up-scaled grid class (F = 384,
// 768, 960; a dozen pairs each) cannot fill the chip with one block per (pair, stripe), so the rows are cut into chunks
// as well; packed 16-bit counters need half the stripes (= half the passes over the samples) of the 32-bit kernel these
// classes used to take.  Partials are added by k_p16_reduce: no atomics, the same sum in every run.
__global__ void __launch_bounds__(1024) k_hist2d_u16_chunks(const Hist2DPair* __restrict__ pairs, int B, int64_t N, int F, int R,
                                                            int nstripes, int nchunks, unsigned int* __restrict__ part,
                                                            int* __restrict__ overflow) {
    extern __shared__ double sh_raw[];
    unsigned int* sh = reinterpret_cast<unsigned int*>(sh_raw);
    __shared__ double red[16];
    int pair, chunk, stripe;
    decode_block(nstripes, nchunks, pair, chunk, stripe);
    if (pair >= B) return;
    const Hist2DPair P = pairs[pair];
    const int row0 = stripe * R;
    const int nwords = (R * F + 1) / 2;
    for (int i = threadIdx.x; i < nwords; i += 1024) sh[i] = 0;
    __syncthreads();
    int64_t per = (N + nchunks - 1) / nchunks;
    per = (per + 7) & ~(int64_t)7;
    const int64_t lo = (int64_t)chunk * per;
    int64_t hi = lo + per;
    if (hi > N) hi = N;
    unsigned int nacc = 0;
    auto visit = [&](unsigned cx, unsigned cy) {
        const unsigned r = cy - (unsigned)row0;
        if (r < (unsigned)R && cx < (unsigned)F && cy < (unsigned)F) {
            const unsigned a = r * (unsigned)F + cx;
            atomicAdd(&sh[a >> 1], 1u << ((a & 1u) * 16u));
            nacc += 1;
        }
    };
    const int64_t hi8 = lo < hi ? lo + ((hi - lo) & ~(int64_t)7) : lo;
    auto visit8 = [&](const uint4& ax, const uint4& ay) {
        const unsigned xs[4] = {ax.x, ax.y, ax.z, ax.w}, ys[4] = {ay.x, ay.y, ay.z, ay.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            visit(xs[q] & 0xFFFFu, ys[q] & 0xFFFFu);
            visit(xs[q] >> 16, ys[q] >> 16);
        }
    };
    // a register ring of DEPTH iterations of loads, as in k_hist2d_u8_pf (round 5: with one pair of loads per lane in flight
    // and ONE block per CU, the kernel paid a memory latency per eight samples)
    constexpr int DEPTH = 3;
    const int64_t nvec = (hi8 - lo) >> 3;  // 16-byte vectors (eight u16 samples) of this chunk
    const int64_t K = (nvec >> 10) / DEPTH * DEPTH;
    const uint4* gx = reinterpret_cast<const uint4*>(P.ix + lo);
    const uint4* gy = reinterpret_cast<const uint4*>(P.iy + lo);
    if (K > 0) {
        uint4 rx[DEPTH], ry[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            rx[d] = gload_u4(gx + d * 1024 + threadIdx.x), ry[d] = gload_u4(gy + d * 1024 + threadIdx.x);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int64_t k = 0; k < K; k += DEPTH) {
            const int64_t kn = (k + DEPTH < K ? k + DEPTH : K - DEPTH) * 1024 + threadIdx.x;  // (past the end: a harmless re-read)
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                visit8(rx[d], ry[d]);
                rx[d] = gload_u4(gx + kn + d * 1024), ry[d] = gload_u4(gy + kn + d * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (int64_t u = K * 1024 + threadIdx.x; u < nvec; u += 1024) visit8(gload_u4(gx + u), gload_u4(gy + u));
    if (threadIdx.x == 0)
        for (int64_t i = hi8; i < hi; ++i) visit(P.ix[i], P.iy[i]);
    __syncthreads();
    unsigned int* dst = part + (((int64_t)pair * nchunks + chunk) * nstripes + stripe) * (int64_t)nwords;
    unsigned int total = 0;
    for (int i = threadIdx.x; i < nwords; i += 1024) {
        const unsigned int v = sh[i];
        total += (v & 0xffffu) + (v >> 16);
        dst[i] = v;
    }
    const double t = block_sum((double)total, red), a = block_sum((double)nacc, red);
    if (threadIdx.x == 0 && t != a) atomicOr(&overflow[pair], 1);
}

// hist[pair][row][col] = sum over chunks of the packed partial counters; grid (word blocks, stripes, B).  A thread owns
// one packed word of a stripe (two neighbouring bins) and keeps eight chunks' loads in flight: the chunks of a word lie
// megabytes apart, so a serial walk (one load latency per chunk) had left this at 50 GB/s on a single up-scaled pair.
__global__ void __launch_bounds__(256) k_p16_reduce(const unsigned int* __restrict__ part, int F, int R, int nstripes, int nchunks,
                                                    double* __restrict__ hist_all) {
    const int pair = blockIdx.z, stripe = blockIdx.y;
    const int nwords = (R * F + 1) / 2;
    const int rows = min(R, F - stripe * R);
    const int wd = blockIdx.x * 256 + threadIdx.x;
    if (2 * wd >= rows * F) return;
    const int64_t step = (int64_t)nstripes * nwords;
    const unsigned int* p = part + ((int64_t)pair * nchunks * nstripes + stripe) * (int64_t)nwords + wd;
    unsigned int lo = 0, hi = 0;
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
        unsigned int v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(p + (c + q) * step);
#pragma unroll
        for (int q = 0; q < 8; ++q) lo += v[q] & 0xffffu, hi += v[q] >> 16;
    }
    for (; c < nchunks; ++c) {
        const unsigned int v = p[c * step];
        lo += v & 0xffffu, hi += v >> 16;
    }
    double* hist = hist_all + (int64_t)pair * F * F + (int64_t)stripe * R * F;
    hist[2 * wd] = (double)lo;  // index in the stripe = row * F + col: the stripe's rows are contiguous in the grid
    if (2 * wd + 1 < rows * F) hist[2 * wd + 1] = (double)hi;
}

// min / max of a*x + b*y over the samples; grid (nblk, B)
__global__ void k_minmax_affine(const Hist2DPair* __restrict__ pairs, int64_t N, double* __restrict__ part) {
    __shared__ double red[16];
    const Hist2DPair P = pairs[blockIdx.y];
    double mn = INFINITY, mx = -INFINITY;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t Ne = N & ~(int64_t)1;
    for (int64_t i = 2 * gtid; i < Ne; i += 2 * gsz) {
        const double2 xv = gload_d2(P.x + i);
        const double2 yv = gload_d2(P.y + i);
        const double p0 = P.r0 * xv.x + P.r1 * yv.x, p1 = P.r0 * xv.y + P.r1 * yv.y;
        mn = fmin(mn, fmin(p0, p1));
        mx = fmax(mx, fmax(p0, p1));
    }
    if (gtid == 0 && Ne < N) {
        const double p0 = P.r0 * P.x[Ne] + P.r1 * P.y[Ne];
        mn = fmin(mn, p0);
        mx = fmax(mx, p0);
    }
    const double r0 = block_min(mn, red), r1 = block_max(mx, red);
    if (threadIdx.x == 0) {
        double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        p[0] = r0, p[1] = r1;
    }
}

// Several sheared pairs share their columns (a correlated block of 5 parameters gives 10 pairs over 5 columns): a group
// of <= 16 pairs over <= 8 distinct columns reads each column ONCE.  Every thread parks its 16-B loads in its own LDS
// slots (LDS as indexable per-thread storage: no barrier) and evaluates the pairs from there.
#define MMG_COLS 8
#define MMG_PAIRS 16
struct MinmaxGroup {
    const double* col[MMG_COLS];
    double r0[MMG_PAIRS], r1[MMG_PAIRS];
    int a[MMG_PAIRS], b[MMG_PAIRS];
    int ncols, npairs;
};

__global__ void __launch_bounds__(256) k_minmax_affine_grouped(const MinmaxGroup* __restrict__ groups, int64_t N,
                                                               double* __restrict__ part) {
    __shared__ double2 slot[MMG_COLS][256];
    __shared__ double red[16];
    const MinmaxGroup& G = groups[blockIdx.y];
    const int ncols = G.ncols, npairs = G.npairs;
    double mn[MMG_PAIRS], mx[MMG_PAIRS];
#pragma unroll
    for (int p = 0; p < MMG_PAIRS; ++p) mn[p] = INFINITY, mx[p] = -INFINITY;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t Ne = N & ~(int64_t)1;
    for (int64_t i = 2 * gtid; i < Ne; i += 2 * gsz) {
        // all MMG_COLS loads unconditionally (slots past ncols re-read the last column: an L1 hit) and all of them requested
        // before the first is stored: predicated on the group's column count, every load sat in its own basic block with its
        // LDS store and was waited for there -- eight memory latencies in sequence per iteration (ISA reading, round 5)
        double2 v[MMG_COLS];
#pragma unroll
        for (int c = 0; c < MMG_COLS; ++c) v[c] = gload_d2(G.col[min(c, ncols - 1)] + i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < MMG_COLS; ++c) slot[c][threadIdx.x] = v[c];
#pragma unroll
        for (int p = 0; p < MMG_PAIRS; ++p)
            if (p < npairs) {
                const double2 xv = slot[G.a[p]][threadIdx.x], yv = slot[G.b[p]][threadIdx.x];
                const double p0 = G.r0[p] * xv.x + G.r1[p] * yv.x, p1 = G.r0[p] * xv.y + G.r1[p] * yv.y;
                mn[p] = fmin(mn[p], fmin(p0, p1));
                mx[p] = fmax(mx[p], fmax(p0, p1));
            }
    }
    if (gtid == 0 && Ne < N) {
#pragma unroll
        for (int p = 0; p < MMG_PAIRS; ++p)
            if (p < npairs) {
                const double p0 = G.r0[p] * G.col[G.a[p]][Ne] + G.r1[p] * G.col[G.b[p]][Ne];
                mn[p] = fmin(mn[p], p0);
                mx[p] = fmax(mx[p], p0);
            }
    }
    double* out = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * MMG_PAIRS * 2;
#pragma unroll
    for (int p = 0; p < MMG_PAIRS; ++p)
        if (p < npairs) {
            const double r0 = block_min(mn[p], red), r1 = block_max(mx[p], red);
            if (threadIdx.x == 0) out[2 * p] = r0, out[2 * p + 1] = r1;
        }
}

// =============================================================================================================
// How many row chunks to cut every (pair, stripe) unit into.  The 128-KB counter table leaves one block per CU, so a
// launch runs in ceil(blocks / CUs) rounds of one block time each, and a block's time is proportional to its rows:
// the cost of c chunks is rounds(c) / c.  (The old rule, "at least two blocks per CU", gave 79 sheared pairs 553
// blocks = 2.16 -> three rounds of N/7 rows; three chunks give one round of N/3.)  A small penalty per chunk accounts
// for the zero-fill, flush and reduction of its partial table; at most 16 chunks.
static int pick_chunks(const gd_ctx* ctx, int64_t units, int64_t rows) {
    int best = 1;
    double best_cost = 1e30;
    for (int c = 1; c <= 16; ++c) {
        if ((int64_t)c * 65536 > rows && c > 1) break;
        const int64_t blocks = units * c;
        const double rounds = (double)((blocks + ctx->cu_count - 1) / ctx->cu_count);
        const double cost = rounds / c + 0.004 * c;
        if (cost < best_cost - 1e-12) best_cost = cost, best = c;
    }
    return best;
}

template <int MODE>
static int launch_hist2d(gd_ctx* ctx, int B, const std::vector<Hist2DPair>& hp, int F, double* d_hist, bool allow_p16 = true);

// unit weights, fp64 columns: packed 16-bit counters, partial grids per chunk reduced without atomics; pairs whose
// counters wrapped are redone with the 32-bit kernel
template <int MODE>
static int launch_hist2d_p16(gd_ctx* ctx, int B, const std::vector<Hist2DPair>& hp, int F, double* d_hist) {
    int R = LDS_HIST_BYTES / (F * 2);
    if (R > F) R = F;
    const int nstripes = (F + R - 1) / R;
    int nchunks = pick_chunks(ctx, (int64_t)B * nstripes, ctx->N);
    const int nwords = (R * F + 1) / 2;
    // Many pairs over few columns (the sheared pairs of a triangle: ten pairs per correlated block of five columns): row
    // TILES of 128 K samples, tile-major, so that a column's megabyte is fetched from HBM once for all the pairs that read
    // it; the price is one 128-KB partial table per (pair, tile) through scratch (6 % of the tile's 2 MB of samples).
    int tile_major = 0;
    // Measured (scripts/r04_shear_order.py, 79 pairs, N = 1e7): 2.60 ms against 2.43 ms pair-major -- the kernel is bound by
    // the loads a lane keeps in flight, not by HBM; kept as an experiment switch.
    if (B >= 16 && nstripes == 1 && ctx->N >= (1 << 20) && getenv("GDHIP_P16_TILE_MAJOR") != nullptr) {
        int64_t tiles = (ctx->N + 131071) / 131072;
        const int64_t cap = ((int64_t)3 << 30) / ((int64_t)B * nwords * 4);  // at most 3 GB of partial tables
        if (tiles > cap) tiles = cap;
        if (tiles > nchunks) nchunks = (int)tiles, tile_major = 1;
    }
    const int units = (B * nchunks + 7) / 8 * 8;
    const int64_t nblocks = (int64_t)units * nstripes;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_pairs = take((int64_t)B * sizeof(Hist2DPair)), o_flags = take((int64_t)B * 4),
                  o_part = take((int64_t)B * nchunks * nstripes * nwords * 4);
    char* base = (char*)gd_scratch2(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    Hist2DPair* d_pairs = (Hist2DPair*)(base + o_pairs);
    int* d_flags = (int*)(base + o_flags);
    unsigned int* d_part = (unsigned int*)(base + o_part);
    GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair)));
    GD_HIP(hipMemsetAsync(d_flags, 0, (size_t)B * 4, ctx->stream));
    GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_f64_p16<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
    k_hist2d_f64_p16<MODE><<<(unsigned)nblocks, 1024, (size_t)nwords * 4, ctx->stream>>>(d_pairs, B, ctx->N, F, R, nstripes,
                                                                                      nchunks, tile_major, d_part, d_flags);
    GD_KERNEL_CHECK();
    k_p16_reduce<<<dim3((unsigned)((nwords + 255) / 256), nstripes, B), 256, 0, ctx->stream>>>(d_part, F, R, nstripes, nchunks, d_hist);
    GD_KERNEL_CHECK();
    std::vector<int> hf((size_t)B);
    GD_TRY(gd_fetch(ctx, hf.data(), d_flags, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    std::vector<int> flagged;
    for (int b = 0; b < B; ++b)
        if (hf[b]) flagged.push_back(b);
    int rc = GD_OK;
    if (!flagged.empty()) {
        const int nf = (int)flagged.size();
        std::vector<Hist2DPair> sub((size_t)nf);
        for (int q = 0; q < nf; ++q) sub[q] = hp[flagged[q]];
        double* tmp = nullptr;
        GD_HIP(hipMalloc((void**)&tmp, (size_t)nf * F * F * 8));
        rc = launch_hist2d<MODE>(ctx, nf, sub, F, tmp, false);
        for (int q = 0; q < nf && rc == GD_OK; ++q)
            if (hipMemcpyAsync(d_hist + (int64_t)flagged[q] * F * F, tmp + (int64_t)q * F * F, (size_t)F * F * 8,
                               hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                rc = gd_fail(ctx, GD_ERR_HIP, "copy of redone histogram failed");
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
    }
    return rc;
}

template <int MODE>
static int launch_hist2d(gd_ctx* ctx, int B, const std::vector<Hist2DPair>& hp, int F, double* d_hist, bool allow_p16) {
    GD_REQUIRE(F >= 2 && F <= 4096, "fine_bins_2D out of range");
    const bool has_w = ctx->w != nullptr;
    if (MODE != 2 && !has_w && allow_p16 && !getenv("GDHIP_NO_P16")) return launch_hist2d_p16<(MODE == 2 ? 0 : MODE)>(ctx, B, hp, F, d_hist);
    const bool u32bins = !has_w || ctx->w_integral;  // integral weights: exact u32 counters, twice the rows per stripe
    const int binbytes = u32bins ? 4 : 8;
    int R = LDS_HIST_BYTES / (F * binbytes);
    GD_REQUIRE(R >= 1, "fine_bins_2D too large for the LDS stripe");
    if (R > F) R = F;
    const int nstripes = (F + R - 1) / R;
    const int nchunks = pick_chunks(ctx, (int64_t)B * nstripes, ctx->N);
    const int units = (B * nchunks + 7) / 8 * 8;
    const int64_t nblocks = (int64_t)units * nstripes;
    Hist2DPair* d_pairs = (Hist2DPair*)gd_scratch2(ctx, (int64_t)B * sizeof(Hist2DPair));
    if (!d_pairs) return GD_ERR_NOMEM;
    GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair)));
    if (nchunks > 1) GD_HIP(hipMemsetAsync(d_hist, 0, (size_t)B * F * F * 8, ctx->stream));
    const size_t lds = (size_t)R * F * binbytes;
    if (has_w && !u32bins) {
        auto kern = k_hist2d<MODE, true, double>;
        GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
        kern<<<(unsigned)nblocks, 1024, lds, ctx->stream>>>(d_pairs, B, ctx->w, ctx->N, F, R, nstripes, nchunks, d_hist);
    } else if (has_w) {
        auto kern = k_hist2d<MODE, true, unsigned int>;
        GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
        kern<<<(unsigned)nblocks, 1024, lds, ctx->stream>>>(d_pairs, B, ctx->w, ctx->N, F, R, nstripes, nchunks, d_hist);
    } else {
        auto kern = k_hist2d<MODE, false, unsigned int>;
        GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
        kern<<<(unsigned)nblocks, 1024, lds, ctx->stream>>>(d_pairs, B, nullptr, ctx->N, F, R, nstripes, nchunks,
                                                            d_hist);
    }
    GD_KERNEL_CHECK();
    GD_TRY(gd_stream_sync(ctx));  // hp / d_pairs lifetime
    return GD_OK;
}

extern "C" {

// out: host ncols x F, or -- with `out_on_device` -- device memory the histograms are left in (no copy, no sync)
static int hist1d_core(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                       double* out, bool out_on_device) {
    GD_REQUIRE(ctx && cols && binmin && width && out && ncols > 0, "bad argument");
    GD_REQUIRE(F >= 2 && F <= 4096, "fine_bins out of range (2..4096)");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    const int nblk = ctx->cu_count;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_part = take((int64_t)ncols * nblk * F * 8), o_out = take((int64_t)ncols * F * 8),
                  o_idx = take((int64_t)ncols * 4), o_b = take((int64_t)ncols * 8), o_w = take((int64_t)ncols * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    double* d_part = (double*)(base + o_part);
    double* d_out = out_on_device ? out : (double*)(base + o_out);
    int32_t* d_idx = (int32_t*)(base + o_idx);
    double* d_b = (double*)(base + o_b);
    double* d_w = (double*)(base + o_w);
    GD_TRY(gd_h2d(ctx, d_idx, cols, (size_t)ncols * 4));
    GD_TRY(gd_h2d(ctx, d_b, binmin, (size_t)ncols * 8));
    GD_TRY(gd_h2d(ctx, d_w, width, (size_t)ncols * 8));
    dim3 grid(nblk, ncols);
    const size_t lds = (size_t)4 * F * 8;
    if (ctx->w) {
        GD_HIP(hipFuncSetAttribute((const void*)k_hist1d<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 8));
        k_hist1d<true><<<grid, 256, lds, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, ctx->w, ctx->N, d_b, d_w, F, d_part);
    } else {
        GD_HIP(hipFuncSetAttribute((const void*)k_hist1d<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 8));
        k_hist1d<false><<<grid, 256, lds, ctx->stream>>>(ctx->cols, ctx->ld, d_idx, nullptr, ctx->N, d_b, d_w, F, d_part);
    }
    GD_KERNEL_CHECK();
    k_hist1d_reduce<<<dim3((F + 255) / 256, ncols), 256, 0, ctx->stream>>>(d_part, nblk, F, d_out);
    GD_KERNEL_CHECK();
    if (out_on_device) return GD_OK;
    GD_TRY(gd_fetch(ctx, out, d_out, (size_t)ncols * F * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_hist1d(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
              double* out) {
    return hist1d_core(ctx, cols, ncols, binmin, width, F, out, false);
}
int gd_hist1d_dev(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                  void* d_out) {
    return hist1d_core(ctx, cols, ncols, binmin, width, F, (double*)d_out, true);
}

int gd_bin_indices(gd_ctx* ctx, int32_t col, double binmin, double width, int32_t round_half, int32_t F,
                   int32_t* idx_out, int64_t* n_out_of_range) {
    GD_REQUIRE(ctx && idx_out, "bad argument");
    GD_REQUIRE(ctx->cols && col >= 0 && col < ctx->n + GD_EXTRA_COLS, "bad column");
    char* base = (char*)gd_scratch(ctx, ctx->N * 4 + 256);
    if (!base) return GD_ERR_NOMEM;
    unsigned long long* d_bad = (unsigned long long*)base;
    int32_t* d_idx = (int32_t*)(base + 256);
    GD_HIP(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    const double* x = ctx->cols + (int64_t)col * ctx->ld;
    if (round_half)
        k_bin_indices<true, int32_t><<<4 * ctx->cu_count, 256, 0, ctx->stream>>>(x, ctx->N, binmin, width, F, d_idx, d_bad);
    else
        k_bin_indices<false, int32_t><<<4 * ctx->cu_count, 256, 0, ctx->stream>>>(x, ctx->N, binmin, width, F, d_idx, d_bad);
    GD_KERNEL_CHECK();
    unsigned long long bad = 0;
    GD_TRY(gd_fetch(ctx, idx_out, d_idx, (size_t)ctx->N * 4));
    GD_TRY(gd_fetch(ctx, &bad, d_bad, 8));
    GD_TRY(gd_stream_sync(ctx));
    if (n_out_of_range) *n_out_of_range = (int64_t)bad;
    return GD_OK;
}

int gd_prebin(gd_ctx* ctx, int32_t col, double binmin, double width, int32_t F, void* d_idx_u16) {
    GD_REQUIRE(ctx && d_idx_u16, "bad argument");
    GD_REQUIRE(ctx->cols && col >= 0 && col < ctx->n + GD_EXTRA_COLS, "bad column");
    GD_REQUIRE(F >= 2 && F < 65535, "F out of range for u16 indices");
    {
        void* const idx1[1] = {d_idx_u16};
        const BucketRoute R = bucket_route(ctx, &col, 1, &binmin, &width, idx1);
        if (!R.recs.empty()) {  // from the column's 2-byte buckets (stream-ordered like the kernel below: no wait here)
            char* base = (char*)gd_scratch2(ctx, bucket_route_bytes(ctx, R));
            if (!base) return GD_ERR_NOMEM;
            unsigned long long* unused = nullptr;
            return bucket_route_launch<false>(ctx, R, base, F, &unused);
        }
    }
    const double* x = ctx->cols + (int64_t)col * ctx->ld;
    k_prebin<<<8 * ctx->cu_count, 256, 0, ctx->stream>>>(x, ctx->N, binmin, width, F, (unsigned short*)d_idx_u16);
    GD_KERNEL_CHECK();
    return GD_OK;
}

int gd_prebin_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                    void* const* d_idx_u16) {
    GD_REQUIRE(ctx && cols && binmin && width && d_idx_u16 && ncols > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(F >= 2 && F < 65535, "F out of range for u16 indices");
    for (int c = 0; c < ncols; ++c)
        GD_REQUIRE(cols[c] >= 0 && cols[c] < ctx->n + GD_EXTRA_COLS && d_idx_u16[c], "bad column / null index buffer");
    const BucketRoute R = bucket_route(ctx, cols, ncols, binmin, width, d_idx_u16);
    std::vector<PrebinCol> hc;
    for (int c = 0; c < ncols; ++c) {
        if (R.slot_of[c] >= 0) continue;
        PrebinCol r;
        r.x = ctx->cols + (int64_t)cols[c] * ctx->ld;
        r.idx = (unsigned short*)d_idx_u16[c];
        r.binmin = binmin[c];
        r.width = width[c];
        hc.push_back(r);
    }
    const int nfp = (int)hc.size();
    const int64_t o_route = ((int64_t)nfp * sizeof(PrebinCol) + 255) / 256 * 256;
    char* base = (char*)gd_scratch2(ctx, o_route + bucket_route_bytes(ctx, R));
    if (!base) return GD_ERR_NOMEM;
    PrebinCol* d_c = (PrebinCol*)base;
    if (nfp) {
        GD_TRY(gd_h2d(ctx, d_c, hc.data(), (size_t)nfp * sizeof(PrebinCol)));
        int nblk = (int)((ctx->N / 8 + 255) / 256);  // one 8-sample iteration per thread at most, like the single-column kernel
        if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
        if (nblk < 1) nblk = 1;
        k_prebin_batch<<<dim3(nblk, nfp), 256, 0, ctx->stream>>>(d_c, ctx->N, F);
        GD_KERNEL_CHECK();
    }
    if (!R.recs.empty()) {
        unsigned long long* unused = nullptr;
        GD_TRY(bucket_route_launch<false>(ctx, R, base + o_route, F, &unused));
    }
    GD_TRY(gd_stream_sync(ctx));  // hc / d_c lifetime
    return GD_OK;
}

int gd_hist2d(gd_ctx* ctx, int32_t B, const int32_t* colx, const int32_t* coly, const double* binminx,
              const double* widthx, const double* binminy, const double* widthy, int32_t F, void* d_hist) {
    GD_REQUIRE(ctx && colx && coly && binminx && widthx && binminy && widthy && d_hist && B > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    std::vector<Hist2DPair> hp((size_t)B);
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(colx[b] >= 0 && colx[b] < ctx->n + GD_EXTRA_COLS && coly[b] >= 0 && coly[b] < ctx->n + GD_EXTRA_COLS, "column out of range");
        Hist2DPair& p = hp[b];
        memset(&p, 0, sizeof p);
        p.x = ctx->cols + (int64_t)colx[b] * ctx->ld;
        p.y = ctx->cols + (int64_t)coly[b] * ctx->ld;
        p.bx = binminx[b], p.wx = widthx[b], p.by = binminy[b], p.wy = widthy[b];
    }
    return launch_hist2d<0>(ctx, B, hp, F, (double*)d_hist);
}

static int launch_hist2d_u16(gd_ctx* ctx, int B, const std::vector<Hist2DPair>& hp, int F, double* d_hist,
                             std::vector<int>& flagged) {
    int R = LDS_HIST_BYTES / (F * 2);
    if (R > F) R = F;
    const int nstripes = (F + R - 1) / R;
    const int units = (B + 7) / 8 * 8;
    const int64_t nblocks = (int64_t)units * nstripes;
    const int64_t o_flags = ((int64_t)B * sizeof(Hist2DPair) + 255) / 256 * 256;
    char* base = (char*)gd_scratch2(ctx, o_flags + (int64_t)B * 4);
    if (!base) return GD_ERR_NOMEM;
    Hist2DPair* d_pairs = (Hist2DPair*)base;
    int* d_flags = (int*)(base + o_flags);
    GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair)));
    GD_HIP(hipMemsetAsync(d_flags, 0, (size_t)B * 4, ctx->stream));
    const size_t lds = ((size_t)R * F + 1) / 2 * 4;
    if (ctx->w8) {
        GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_u16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
        k_hist2d_u16<true><<<(unsigned)nblocks, 1024, lds, ctx->stream>>>(d_pairs, B, ctx->N, F, R, nstripes, ctx->w8, d_hist,
                                                                         d_flags);
    } else {
        GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_u16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
        k_hist2d_u16<false><<<(unsigned)nblocks, 1024, lds, ctx->stream>>>(d_pairs, B, ctx->N, F, R, nstripes, nullptr, d_hist,
                                                                          d_flags);
    }
    GD_KERNEL_CHECK();
    std::vector<int> hf((size_t)B);
    GD_TRY(gd_fetch(ctx, hf.data(), d_flags, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    flagged.clear();
    for (int b = 0; b < B; ++b)
        if (hf[b]) flagged.push_back(b);
    return GD_OK;
}

int gd_hist2d_prebinned(gd_ctx* ctx, int32_t B, const void* const* d_idx_x, const void* const* d_idx_y, int32_t F,
                        void* d_hist) {
    GD_REQUIRE(ctx && d_idx_x && d_idx_y && d_hist && B > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(F >= 2 && F <= 4096, "fine_bins_2D out of range");
    std::vector<Hist2DPair> hp((size_t)B);
    for (int b = 0; b < B; ++b) {
        Hist2DPair& p = hp[b];
        memset(&p, 0, sizeof p);
        p.ix = (const unsigned short*)d_idx_x[b];
        p.iy = (const unsigned short*)d_idx_y[b];
        GD_REQUIRE(p.ix && p.iy, "null index column");
    }
    {
        int R16 = LDS_HIST_BYTES / (F * 2);
        if (R16 > F) R16 = F;
        const int nstripes16 = (F + R16 - 1) / R16;
        if (!ctx->w && (int64_t)B * nstripes16 < ctx->cu_count && !getenv("GDHIP_NO_U16_CHUNKS")) {
            // few pairs (the up-scaled grid classes): packed counters over (pair, chunk of rows, stripe) blocks, partials
            // reduced without atomics; pairs whose counters wrapped are redone with the 32-bit kernel
            const int R = R16, nstripes = nstripes16;
            const int nchunks = pick_chunks(ctx, (int64_t)B * nstripes, ctx->N);
            const int units = (B * nchunks + 7) / 8 * 8;
            const int64_t nblocks = (int64_t)units * nstripes;
            const int nwords = (R * F + 1) / 2;
            const int64_t o_flags = ((int64_t)B * sizeof(Hist2DPair) + 255) / 256 * 256, o_part = o_flags + ((int64_t)B * 4 + 255) / 256 * 256;
            char* base = (char*)gd_scratch2(ctx, o_part + (int64_t)B * nchunks * nstripes * nwords * 4);
            if (!base) return GD_ERR_NOMEM;
            Hist2DPair* d_pairs = (Hist2DPair*)base;
            int* d_flags = (int*)(base + o_flags);
            unsigned int* d_part = (unsigned int*)(base + o_part);
            GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair)));
            GD_HIP(hipMemsetAsync(d_flags, 0, (size_t)B * 4, ctx->stream));
            GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_u16_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
            k_hist2d_u16_chunks<<<(unsigned)nblocks, 1024, (size_t)nwords * 4, ctx->stream>>>(d_pairs, B, ctx->N, F, R, nstripes, nchunks,
                                                                                              d_part, d_flags);
            GD_KERNEL_CHECK();
            k_p16_reduce<<<dim3((unsigned)((nwords + 255) / 256), nstripes, B), 256, 0, ctx->stream>>>(d_part, F, R, nstripes, nchunks, (double*)d_hist);
            GD_KERNEL_CHECK();
            std::vector<int> hf((size_t)B);
            GD_TRY(gd_fetch(ctx, hf.data(), d_flags, (size_t)B * 4));
            GD_TRY(gd_stream_sync(ctx));
            std::vector<int> flagged;
            for (int b = 0; b < B; ++b)
                if (hf[b]) flagged.push_back(b);
            int rc = GD_OK;
            if (!flagged.empty()) {
                const int nf = (int)flagged.size();
                std::vector<Hist2DPair> sub((size_t)nf);
                for (int q = 0; q < nf; ++q) sub[q] = hp[flagged[q]];
                double* tmp = nullptr;
                GD_HIP(hipMalloc((void**)&tmp, (size_t)nf * F * F * 8));
                rc = launch_hist2d<2>(ctx, nf, sub, F, tmp);
                for (int q = 0; q < nf && rc == GD_OK; ++q)
                    if (hipMemcpyAsync((double*)d_hist + (int64_t)flagged[q] * F * F, tmp + (int64_t)q * F * F, (size_t)F * F * 8,
                                       hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                        rc = gd_fail(ctx, GD_ERR_HIP, "copy of redone histogram failed");
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(tmp);
            }
            return rc;
        }
        // (real weights: four stripe passes with ds_add_f64.  Round 5 measured a one-pass alternative -- samples partitioned by
        // stripe once per y column into 16-byte records (row, row in stripe, weight), a (pair, stripe) block walking only its
        // quarter -- at 43.6 ms per 1225 pairs against 41.3 ms here: 16 bytes per (sample, pair) make it bound by the L2-miss
        // path (80 GB per 400-pair launch at 0.74 of the peak; with the pairs of a y column on one XCD 42 GB, but 19 ms).  The
        // roof of the adds themselves is 7.8 ms (2.5 ds_add_f64 lanes per clock and CU, scripts/micro/lds_atomic_f64_roof.hip);
        // what keeps both forms away from it is the 12-16 bytes every (sample, pair) visit must fetch.  Not kept; DESIGN.md.)
        // the 16-bit kernel gives each (pair, stripe) to ONE block: only worth it when that fills the chip
        if ((ctx->w && !ctx->w8) || (int64_t)B * nstripes16 < ctx->cu_count)
            return launch_hist2d<2>(ctx, B, hp, F, (double*)d_hist);
    }
    // unit weights or byte multiplicities, batched: 16-bit packed LDS counters, exact overflow detection, 32-bit redo
    // for flagged pairs
    std::vector<int> flagged;
    int rc = launch_hist2d_u16(ctx, B, hp, F, (double*)d_hist, flagged);
    if (rc) return rc;
    if (!flagged.empty()) {
        const int nf = (int)flagged.size();
        std::vector<Hist2DPair> sub((size_t)nf);
        for (int q = 0; q < nf; ++q) sub[q] = hp[flagged[q]];
        double* tmp = nullptr;
        GD_HIP(hipMalloc((void**)&tmp, (size_t)nf * F * F * 8));
        rc = launch_hist2d<2>(ctx, nf, sub, F, tmp);
        if (rc == GD_OK)
            for (int q = 0; q < nf && rc == GD_OK; ++q)
                if (hipMemcpyAsync((double*)d_hist + (int64_t)flagged[q] * F * F, tmp + (int64_t)q * F * F, (size_t)F * F * 8,
                                   hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                    rc = gd_fail(ctx, GD_ERR_HIP, "copy of redone histogram failed");
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
    }
    return rc;
}

int gd_prebin8_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                     void* const* d_idx_out, int64_t* bad_out) {
    GD_REQUIRE(ctx && cols && binmin && width && d_idx_out && bad_out && ncols > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(F >= 2 && F <= 256, "byte indices need fine_bins_2D <= 256");
    for (int c = 0; c < ncols; ++c) GD_REQUIRE(cols[c] >= 0 && cols[c] < ctx->n + GD_EXTRA_COLS && d_idx_out[c], "bad column");
    // columns whose bucket columns are valid (the quantile select has walked them) are binned from those 2 bytes per sample
    const BucketRoute R = bucket_route(ctx, cols, ncols, binmin, width, d_idx_out);
    std::vector<PrebinCol8> hc;
    std::vector<int> fp_of((size_t)ncols, -1);
    for (int c = 0; c < ncols; ++c) {
        if (R.slot_of[c] >= 0) continue;
        PrebinCol8 r;
        r.x = ctx->cols + (int64_t)cols[c] * ctx->ld;
        r.idx = (unsigned char*)d_idx_out[c];
        r.binmin = binmin[c];
        r.width = width[c];
        fp_of[c] = (int)hc.size();
        hc.push_back(r);
    }
    const int nfp = (int)hc.size();
    const int64_t o_bad = ((int64_t)nfp * sizeof(PrebinCol8) + 255) / 256 * 256;
    const int64_t o_route = o_bad + ((int64_t)nfp * 8 + 255) / 256 * 256;
    char* base = (char*)gd_scratch2(ctx, o_route + bucket_route_bytes(ctx, R));
    if (!base) return GD_ERR_NOMEM;
    PrebinCol8* d_c = (PrebinCol8*)base;
    unsigned long long* d_bad = (unsigned long long*)(base + o_bad);
    if (nfp) {
        GD_TRY(gd_h2d(ctx, d_c, hc.data(), (size_t)nfp * sizeof(PrebinCol8)));
        GD_HIP(hipMemsetAsync(d_bad, 0, (size_t)nfp * 8, ctx->stream));
        int nblk = (int)((ctx->N / 8 + 255) / 256);
        if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
        if (nblk < 1) nblk = 1;
        k_prebin8_batch<<<dim3(nblk, nfp), 256, 0, ctx->stream>>>(d_c, ctx->N, F, d_bad);
        GD_KERNEL_CHECK();
    }
    unsigned long long* d_bad_bq = nullptr;
    if (!R.recs.empty()) GD_TRY(bucket_route_launch<true>(ctx, R, base + o_route, F, &d_bad_bq));
    std::vector<unsigned long long> hb((size_t)nfp), hq(R.recs.size());
    if (nfp) GD_TRY(gd_fetch(ctx, hb.data(), d_bad, (size_t)nfp * 8));
    if (!R.recs.empty()) GD_TRY(gd_fetch(ctx, hq.data(), d_bad_bq, hq.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int c = 0; c < ncols; ++c) bad_out[c] = (int64_t)(R.slot_of[c] >= 0 ? hq[R.slot_of[c]] : hb[fp_of[c]]);
    return GD_OK;
}

// ---- real weights, F = 256: samples partitioned by stripe once per y column (round 6) ------------------------------------
// A pair's fp64 (here: 64-bit fixed point) table is 512 KB; a CU's LDS holds 128 KB.  The four-pass kernel
// (k_hist2d<.., true>) lets four 64-row stripe blocks each read ALL samples -- index bytes and the 8-byte weight four times per
// (sample, pair), 40+ bytes and ~40 instructions -- and ran at 0.19 of the atomics' roof.  Here the samples are PARTITIONED BY
// STRIPE once per y column ("key": the pairs of a triangle share their y columns) by a streaming counting sort, into WS = 16
// stripes of 16 rows, so that ONE block holds the 32-KB stripe tables of FOUR pairs of the key and feeds all four from one
// read of the stripe's weights (measured first with 4 stripes x 1 pair per block: the 8-byte weight per (sample, pair) came
// from HBM -- blocks meant to share it through their XCD's L2 drift apart -- and bound the kernel at 4.9 TB/s):
//   k_wpart_count   per (key, tile of 4096 rows): how many rows fall in each stripe of the key's index column
//   k_wpart_scan    per key: the stripe-major layout -- every (tile, stripe) run padded to 16 entries, so that all later
//                   accesses are aligned 16-byte vectors; pad entries carry weight 0 (they add 0 to bin 0)
//   k_wpart_scatter per (key, tile): a slot per row (LDS cursors; the order inside a run is immaterial: the sums are
//                   integers), then the key's row-in-stripe bytes and FIXED-POINT weights (round(w 2^k), 2^k sum(w) < 2^62:
//                   exact in any order, reruns bit-equal, and ds_add_u64 retires at twice the rate of ds_add_f64) and, for
//                   EVERY x column paired with the key, eight at a time, that column's index bytes in the same order --
//                   sequential reads, run-wise sequential writes, no gathers
//   k_hist2d_wsorted per (key, stripe, group of <= 4 pairs): walks only its sixteenth of the key's stream: per entry 4 x 1
//                   index bytes + 1 + 8 shared bytes, four ds_add_u64; units of one (key, stripe) get block ids on one XCD.
#define WP_TILE 4096
#ifndef WS
#define WS 16          // stripes
#endif
#define WR (256 / WS)  // rows per stripe
#define WPG (WS / 4)   // pairs per block: WPG tables of WR x 256 u64 fill the 128-KB LDS
#define WXB 8          // x columns per staging round of the scatter kernel
struct WKey {
    const unsigned char* iy;   // the key column's byte indices (N)
    unsigned char* ij_perm;    // row within the stripe, partitioned order (npad)
    unsigned long long* wq_perm;  // fixed-point weights, partitioned order (npad)
    int64_t npad;              // entries of the partitioned stream
    int x_first, x_count;      // this key's x columns: entries [x_first, x_first + x_count) of the WX table
};
struct WX {
    const unsigned char* ix;   // an x column's byte indices (N)
    unsigned char* ix_perm;    // the same in the key's partitioned order (npad)
};

__global__ void __launch_bounds__(256) k_wpart_count(const WKey* __restrict__ keys, int64_t N, int ntiles, int* __restrict__ counts) {
    __shared__ int c[WS];
    const WKey K = keys[blockIdx.y];
    if (threadIdx.x < WS) c[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * WP_TILE;
    for (int e = threadIdx.x * 16; e < WP_TILE; e += 256 * 16) {
        const int64_t r = base + e;
        if (r + 16 <= N) {
            const uint4 v = gload_u4(K.iy + r);
            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 16; ++q) atomicAdd(&c[((wd[q >> 2] >> (8 * (q & 3))) & 0xff) / WR], 1);
        } else {
            for (int q = 0; q < 16 && r + q < N; ++q) atomicAdd(&c[K.iy[r + q] / WR], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < WS) counts[((int64_t)blockIdx.y * ntiles + blockIdx.x) * WS + threadIdx.x] = c[threadIdx.x];
}

// one block per key: offsets[key][tile][s] = start of the (tile, s) run in the partitioned stream (entries), stripe-major;
// starts[key][0..WS] = where the stripes begin / the stream ends.  One wave per stripe walks the tiles (lane-strided sums,
// then a serial pass of lane 0 over 64-tile blocks would be shorter still; the kernel is < 1 % of the call).
#define WLPS (1024 / WS)  // threads per stripe in the scan
__global__ void __launch_bounds__(1024) k_wpart_scan(const int* __restrict__ counts, int ntiles, int64_t* __restrict__ offsets,
                                                     int64_t* __restrict__ starts) {
    __shared__ int64_t tot[WS];
    __shared__ int64_t chunk_base[WS][WLPS + 1];
    const int s = threadIdx.x / WLPS, lane = threadIdx.x % WLPS;
    const int* c = counts + (int64_t)blockIdx.x * ntiles * WS;
    // thread l of a stripe owns the tiles [l * per, (l + 1) * per)
    const int per = (ntiles + WLPS - 1) / WLPS;
    const int t0 = lane * per, t1 = min(ntiles, t0 + per);
    int64_t mine = 0;
    for (int i = t0; i < t1; ++i) mine += (c[(int64_t)i * WS + s] + 15) & ~15;
    chunk_base[s][lane] = mine;
    __syncthreads();
    if (lane == 0) {
        int64_t run = 0;
        for (int l = 0; l < WLPS; ++l) {
            const int64_t v = chunk_base[s][l];
            chunk_base[s][l] = run;
            run += v;
        }
        tot[s] = run;
    }
    __syncthreads();
    int64_t off = 0;
    for (int q = 0; q < s; ++q) off += tot[q];
    if (lane == 0) {
        starts[(int64_t)blockIdx.x * (WS + 1) + s] = off;
        if (s == WS - 1) starts[(int64_t)blockIdx.x * (WS + 1) + WS] = off + tot[s];
    }
    off += chunk_base[s][lane];
    int64_t* o = offsets + (int64_t)blockIdx.x * ntiles * WS;
    for (int i = t0; i < t1; ++i) {
        o[(int64_t)i * WS + s] = off;
        off += (c[(int64_t)i * WS + s] + 15) & ~15;
    }
}

// grid (ntiles, nkeys), 1024 threads x 4 rows.
__global__ void __launch_bounds__(1024) k_wpart_scatter(const WKey* __restrict__ keys, const WX* __restrict__ xs, const double* __restrict__ w,
                                                        int64_t N, int ntiles, const int* __restrict__ counts,
                                                        const int64_t* __restrict__ offsets, double wscale) {
    constexpr int STG = WP_TILE + WS * 16;  // the tile's padded runs
    __shared__ unsigned long long stage_w[STG];
    __shared__ unsigned char stage_b[WXB][STG];
    __shared__ int run_start[WS + 1], cursor[WS];
    __shared__ int64_t run_off[WS];
    const WKey K = keys[blockIdx.y];
    const int tile = blockIdx.x;
    const int64_t base = (int64_t)tile * WP_TILE;
    const int tid = threadIdx.x;
    const int* cnt = counts + ((int64_t)blockIdx.y * ntiles + tile) * WS;
    if (tid < WS) cursor[tid] = 0, run_off[tid] = offsets[((int64_t)blockIdx.y * ntiles + tile) * WS + tid];
    if (tid == 0) {
        int a = 0;
        for (int s = 0; s < WS; ++s) {
            run_start[s] = a;
            a += (cnt[s] + 15) & ~15;
        }
        run_start[WS] = a;
    }
    __syncthreads();
    const int total = run_start[WS];  // staged entries of the tile (a multiple of 16)
    for (int e = tid; e < total; e += 1024) stage_w[e] = 0ull;
    for (int e = tid; e < WXB * STG / 4; e += 1024) reinterpret_cast<unsigned int*>(&stage_b[0][0])[e] = 0u;
    // the thread's four consecutive rows, their stripes and slots
    const int64_t r0 = base + 4 * tid;
    unsigned iy4 = 0;
    if (r0 + 4 <= N)
        iy4 = *reinterpret_cast<const unsigned int*>(K.iy + r0);
    else
        for (int q = 0; q < 4; ++q)
            if (r0 + q < N) iy4 |= (unsigned)K.iy[r0 + q] << (8 * q);
    __syncthreads();
    int pos[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        pos[q] = -1;
        if (r0 + q < N) {
            const int st = (int)((iy4 >> (8 * q)) & 0xff) / WR;
            pos[q] = run_start[st] + atomicAdd(&cursor[st], 1);
        }
    }
    // ---- the key's own streams: row in stripe, fixed-point weight
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (pos[q] >= 0) {
            stage_b[0][pos[q]] = (unsigned char)(((iy4 >> (8 * q)) & 0xff) % WR);
            stage_w[pos[q]] = __double2ull_rn(w[r0 + q] * wscale);
        }
    __syncthreads();
    // a staged run s goes to [run_off[s], + padded count) of the stream: 16-byte vectors throughout
    auto stripe_of = [&](int e) {
        int s = 0;
#pragma unroll
        for (int q = WS / 2; q > 0; q >>= 1)
            if (e >= run_start[s + q]) s += q;
        return s;
    };
    for (int v = tid; v < total / 16; v += 1024) {
        const int e = v * 16, s = stripe_of(e);
        *reinterpret_cast<uint4*>(K.ij_perm + run_off[s] + (e - run_start[s])) = *reinterpret_cast<const uint4*>(&stage_b[0][e]);
    }
    for (int v = tid; v < total / 2; v += 1024) {
        const int e = v * 2, s = stripe_of(e);
        *reinterpret_cast<uint4*>(K.wq_perm + run_off[s] + (e - run_start[s])) = *reinterpret_cast<const uint4*>(stage_w + e);
    }
    // ---- every x column paired with the key, WXB at a time, in the same order
    for (int x0 = 0; x0 < K.x_count; x0 += WXB) {
        const int nx = min(WXB, K.x_count - x0);
        unsigned ix4[WXB];
#pragma unroll
        for (int u = 0; u < WXB; ++u) {
            const WX X = xs[K.x_first + x0 + (u < nx ? u : 0)];
            ix4[u] = 0;
            if (r0 + 4 <= N)
                ix4[u] = *reinterpret_cast<const unsigned int*>(X.ix + r0);
            else
                for (int q = 0; q < 4; ++q)
                    if (r0 + q < N) ix4[u] |= (unsigned)X.ix[r0 + q] << (8 * q);
        }
        __syncthreads();  // the previous flush has read the staging area (pad bytes stay 0: they are never written)
#pragma unroll
        for (int u = 0; u < WXB; ++u)
            if (u < nx)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (pos[q] >= 0) stage_b[u][pos[q]] = (unsigned char)((ix4[u] >> (8 * q)) & 0xff);
        __syncthreads();
        for (int v = tid; v < nx * (total / 16); v += 1024) {
            const int u = v / (total / 16), e = (v % (total / 16)) * 16, s = stripe_of(e);
            unsigned char* dst = xs[K.x_first + x0 + u].ix_perm;
            *reinterpret_cast<uint4*>(dst + run_off[s] + (e - run_start[s])) = *reinterpret_cast<const uint4*>(&stage_b[u][e]);
        }
    }
}

struct WUnit {  // one block of k_hist2d_wsorted: stripe `stripe` of up to WPG pairs of key `key`
    int pair[WPG];  // -1: unused slot; pair[0] < 0: nothing to do (padding of an XCD's list)
    int stripe, key;
};

__global__ void __launch_bounds__(1024) k_hist2d_wsorted(const WUnit* __restrict__ units, const WKey* __restrict__ keys,
                                                         const unsigned char* const* __restrict__ ix_perm_of_pair,
                                                         const int64_t* __restrict__ starts, double inv_wscale, double* __restrict__ hist) {
    extern __shared__ double wsorted_sh[];
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(wsorted_sh);  // WPG tables of WR rows x 256 columns
    const WUnit U = units[blockIdx.x];
    if (U.pair[0] < 0) return;
    for (int i = threadIdx.x; i < WPG * WR * 256; i += 1024) tab[i] = 0ull;
    __syncthreads();
    const WKey K = keys[U.key];
    const unsigned char* ixp[WPG];
#pragma unroll
    for (int p = 0; p < WPG; ++p) ixp[p] = ix_perm_of_pair[U.pair[p] >= 0 ? U.pair[p] : U.pair[0]];
    const int64_t a = starts[(int64_t)U.key * (WS + 1) + U.stripe], b = starts[(int64_t)U.key * (WS + 1) + U.stripe + 1];  // multiples of 16
    // (measured and not kept: a second register set with the next 16 entries' loads in flight under the adds -- 11.0 against
    // 10.6 ms per 1225 pairs; 32 stripes x 8 pairs per block -- the adds 8.4 ms, the partitioning 9.9 instead of 8.4: the
    // kernel sits between the ds_add_u64 rate (3.9 ms for the triangle's 1.2e10 adds at random targets, more with a Gaussian's
    // conflicts) and the 39 GB of streams it reads once)
    for (int64_t e = a + 16 * (int64_t)threadIdx.x; e < b; e += 16 * 1024) {
        uint4 vx[WPG], vw[8];
#pragma unroll
        for (int p = 0; p < WPG; ++p) vx[p] = gload_u4(ixp[p] + e);
        const uint4 vy = gload_u4(K.ij_perm + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) vw[q] = gload_u4(K.wq_perm + e + 2 * q);
        const unsigned wy[4] = {vy.x, vy.y, vy.z, vy.w};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned cy = (wy[q >> 2] >> (8 * (q & 3))) & 0xff;
            const uint4 pw = vw[q >> 1];
            const unsigned long long wq = (q & 1) ? ((unsigned long long)pw.w << 32 | pw.z) : ((unsigned long long)pw.y << 32 | pw.x);
#pragma unroll
            for (int p = 0; p < WPG; ++p) {
                const unsigned wx[4] = {vx[p].x, vx[p].y, vx[p].z, vx[p].w};
                const unsigned cx = (wx[q >> 2] >> (8 * (q & 3))) & 0xff;
                if (U.pair[p] >= 0) atomicAdd(&tab[p * WR * 256 + ((cy << 8) | cx)], wq);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < WPG; ++p) {
        if (U.pair[p] < 0) continue;
        double* out = hist + (int64_t)U.pair[p] * 65536 + (int64_t)U.stripe * WR * 256;
        for (int i = threadIdx.x; i < WR * 256; i += 1024) out[i] = (double)tab[p * WR * 256 + i] * inv_wscale;
    }
}

// host side: B pairs of byte index columns, real weights with a known total.  Keys are processed in groups that fit the
// scratch budget; returns GD_OK or an error.  The caller has checked F = 256, ctx->w, ctx->w_sum.
static int hist2d_weighted_sorted(gd_ctx* ctx, int B, const void* const* d_idx_x, const void* const* d_idx_y, double* d_hist) {
    const int64_t N = ctx->N;
    const int ntiles = (int)((N + WP_TILE - 1) / WP_TILE);
    const int64_t npad_max = N + (int64_t)ntiles * WS * 15 + 64;
    const double wscale = ldexp(1.0, 61 - ilogb(ctx->w_sum));
    // group the pairs by their y column
    std::map<const void*, std::vector<int>> by_key;
    std::vector<const void*> key_order;
    for (int b = 0; b < B; ++b) {
        if (!by_key.count(d_idx_y[b])) key_order.push_back(d_idx_y[b]);
        by_key[d_idx_y[b]].push_back(b);
    }
    double budget = 20e9;
    if (const char* e = getenv("GDHIP_WSORT_BYTES")) budget = atof(e);
    size_t k0 = 0;
    while (k0 < key_order.size()) {
        // keys [k0, k1): as many as fit
        size_t k1 = k0;
        int64_t bytes = 0;
        int npairs = 0;
        while (k1 < key_order.size()) {
            const int np = (int)by_key[key_order[k1]].size();
            const int64_t add = npad_max * 9 + (int64_t)np * npad_max;
            if (k1 > k0 && bytes + add > budget) break;
            bytes += add, npairs += np, ++k1;
        }
        const int nk = (int)(k1 - k0);
        // units: (key, stripe) groups on one XCD each, consecutive there (block id = 8 position + XCD)
        std::vector<std::vector<WUnit>> lane(8);
        {
            std::vector<int64_t> load(8, 0);
            for (int k = 0; k < nk; ++k) {
                const std::vector<int>& prs = by_key[key_order[k0 + k]];
                for (int s = 0; s < WS; ++s) {
                    int best = 0;
                    for (int x = 1; x < 8; ++x)
                        if (load[x] < load[best]) best = x;
                    for (size_t q = 0; q < prs.size(); q += WPG) {
                        WUnit u;
                        for (int p = 0; p < WPG; ++p) u.pair[p] = q + p < prs.size() ? prs[q + p] : -1;
                        u.stripe = s, u.key = k;
                        lane[best].push_back(u);
                    }
                    load[best] += (int64_t)prs.size();
                }
            }
        }
        size_t longest = 0;
        for (auto& l : lane) longest = std::max(longest, l.size());
        WUnit none;
        for (int p = 0; p < WPG; ++p) none.pair[p] = -1;
        none.stripe = none.key = 0;
        std::vector<WUnit> units(longest * 8, none);
        for (int x = 0; x < 8; ++x)
            for (size_t q = 0; q < lane[x].size(); ++q) units[q * 8 + x] = lane[x][q];
        std::vector<WKey> hk((size_t)nk);
        std::vector<WX> hx;
        std::vector<const unsigned char*> ixp((size_t)B, nullptr);
        int64_t off = 0;
        auto take = [&](int64_t nbytes) {
            int64_t o = off;
            off += (nbytes + 255) / 256 * 256;
            return o;
        };
        const int64_t o_keys = take((int64_t)nk * sizeof(WKey)), o_xs = take((int64_t)npairs * sizeof(WX)),
                      o_counts = take((int64_t)nk * ntiles * WS * 4), o_offsets = take((int64_t)nk * ntiles * WS * 8),
                      o_starts = take((int64_t)nk * (WS + 1) * 8), o_ixp = take((int64_t)B * 8),
                      o_units = take((int64_t)units.size() * sizeof(WUnit));
        std::vector<int64_t> o_ij((size_t)nk), o_wq((size_t)nk);
        std::vector<int64_t> o_ix;
        for (int k = 0; k < nk; ++k) o_ij[k] = take(npad_max), o_wq[k] = take(npad_max * 8);
        for (int k = 0; k < nk; ++k)
            for (size_t q = 0; q < by_key[key_order[k0 + k]].size(); ++q) o_ix.push_back(take(npad_max));
        char* base = (char*)gd_scratch2(ctx, off);
        if (!base) return GD_ERR_NOMEM;
        int xi = 0;
        for (int k = 0; k < nk; ++k) {
            const std::vector<int>& prs = by_key[key_order[k0 + k]];
            WKey& K = hk[k];
            K.iy = (const unsigned char*)key_order[k0 + k];
            K.ij_perm = (unsigned char*)(base + o_ij[k]);
            K.wq_perm = (unsigned long long*)(base + o_wq[k]);
            K.npad = npad_max;
            K.x_first = xi, K.x_count = (int)prs.size();
            for (int b : prs) {
                WX X;
                X.ix = (const unsigned char*)d_idx_x[b];
                X.ix_perm = (unsigned char*)(base + o_ix[xi]);
                ixp[b] = X.ix_perm;
                hx.push_back(X);
                ++xi;
            }
        }
        WKey* d_keys = (WKey*)(base + o_keys);
        WX* d_xs = (WX*)(base + o_xs);
        int* d_counts = (int*)(base + o_counts);
        int64_t* d_offsets = (int64_t*)(base + o_offsets);
        int64_t* d_starts = (int64_t*)(base + o_starts);
        const unsigned char** d_ixp = (const unsigned char**)(base + o_ixp);
        WUnit* d_units = (WUnit*)(base + o_units);
        GD_TRY(gd_h2d(ctx, d_keys, hk.data(), hk.size() * sizeof(WKey)));
        GD_TRY(gd_h2d(ctx, d_xs, hx.data(), hx.size() * sizeof(WX)));
        GD_TRY(gd_h2d(ctx, d_ixp, ixp.data(), ixp.size() * 8));
        GD_TRY(gd_h2d(ctx, d_units, units.data(), units.size() * sizeof(WUnit)));
        k_wpart_count<<<dim3(ntiles, nk), 256, 0, ctx->stream>>>(d_keys, N, ntiles, d_counts);
        GD_KERNEL_CHECK();
        k_wpart_scan<<<nk, 1024, 0, ctx->stream>>>(d_counts, ntiles, d_offsets, d_starts);
        GD_KERNEL_CHECK();
        k_wpart_scatter<<<dim3(ntiles, nk), 1024, 0, ctx->stream>>>(d_keys, d_xs, ctx->w, N, ntiles, d_counts, d_offsets, wscale);
        GD_KERNEL_CHECK();
        GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_wsorted, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES));
        k_hist2d_wsorted<<<(unsigned)units.size(), 1024, LDS_HIST_BYTES, ctx->stream>>>(d_units, d_keys, d_ixp, d_starts, 1.0 / wscale, d_hist);
        GD_KERNEL_CHECK();
        GD_TRY(gd_stream_sync(ctx));  // (the host tables above; the next group reuses the scratch)
        k0 = k1;
    }
    return GD_OK;
}

int gd_hist2d_prebinned8(gd_ctx* ctx, int32_t B, const void* const* d_idx_x, const void* const* d_idx_y, void* d_hist) {
    GD_REQUIRE(ctx && d_idx_x && d_idx_y && d_hist && B > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    if (ctx->w) {  // real weights: partitioned by stripe once per y column, one 10-byte visit per (sample, pair)
        GD_REQUIRE(!ctx->w8 && ctx->w_sum > 1e-200 && ctx->w_sum < 1e200,
                   "byte-index binning takes unit weights or real weights with a known total (multiplicities: gd_hist2d_prebinned)");
        for (int b = 0; b < B; ++b) GD_REQUIRE(d_idx_x[b] && d_idx_y[b], "null index column");
        return hist2d_weighted_sorted(ctx, B, d_idx_x, d_idx_y, (double*)d_hist);
    }
    const int F = 256;
    std::vector<Hist2DPair8> hp((size_t)B);
    for (int b = 0; b < B; ++b) {
        hp[b].ix = (const unsigned char*)d_idx_x[b];
        hp[b].iy = (const unsigned char*)d_idx_y[b];
        GD_REQUIRE(hp[b].ix && hp[b].iy, "null index column");
    }
    const int nchunks = u8_chunks(ctx, B, ctx->N);
    const int64_t o_flags = ((int64_t)B * sizeof(Hist2DPair8) + 255) / 256 * 256, o_part = o_flags + ((int64_t)B * 4 + 255) / 256 * 256;
    char* base = (char*)gd_scratch2(ctx, o_part + (nchunks > 1 ? (int64_t)B * nchunks * 32768 * 4 : 0));
    if (!base) return GD_ERR_NOMEM;
    Hist2DPair8* d_pairs = (Hist2DPair8*)base;
    int* d_flags = (int*)(base + o_flags);
    GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair8)));
    GD_HIP(hipMemsetAsync(d_flags, 0, (size_t)B * 4, ctx->stream));
    const int nblocks = (B * nchunks + 7) / 8 * 8;
    GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_u8_pf<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES + 128));
    k_hist2d_u8_pf<3><<<nblocks, 1024, LDS_HIST_BYTES + 128, ctx->stream>>>(d_pairs, B, ctx->N, (double*)d_hist, d_flags, nchunks,
                                                                              (unsigned int*)(base + o_part));
    GD_KERNEL_CHECK();
    if (nchunks > 1) {
        k_p8_reduce<<<dim3(32, B), 256, 0, ctx->stream>>>((const unsigned int*)(base + o_part), nchunks, (double*)d_hist);
        GD_KERNEL_CHECK();
    }
    std::vector<int> hf((size_t)B);
    GD_TRY(gd_fetch(ctx, hf.data(), d_flags, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    int rc = GD_OK;
    for (int b = 0; b < B && rc == GD_OK; ++b)
        if (hf[b]) rc = gd_fail(ctx, GD_ERR_SOLVER, "16-bit bin counter wrapped in pair %d: redo with gd_hist2d_prebinned", b);
    return rc;
}

// gd_prebin8_hist2d: gd_prebin8_batch (for the `ncols` stale columns) and gd_hist2d_prebinned8 in ONE stream-ordered
// sequence with ONE wait at the end: the out-of-range counts and the wrap flags come back together.  Between the two
// kernels the host would otherwise wait for a 400-byte fetch that queues behind the previous call's result copies on PCIe.
int gd_prebin8_hist2d(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width,
                      void* const* d_idx_out, int64_t* bad_out, int32_t B, const void* const* d_idx_x, const void* const* d_idx_y,
                      void* d_hist) {
    GD_REQUIRE(ctx && d_idx_x && d_idx_y && d_hist && B > 0 && ncols >= 0, "bad argument");
    GD_REQUIRE(ncols == 0 || (cols && binmin && width && d_idx_out && bad_out), "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    if (ctx->w) {  // real weights: the two steps one after the other (the weighted histograms are several launches anyway)
        if (ncols) {
            GD_TRY(gd_prebin8_batch(ctx, cols, ncols, binmin, width, 256, d_idx_out, bad_out));
            int64_t nbad = 0;
            for (int c = 0; c < ncols; ++c) nbad += bad_out[c];
            if (nbad) return gd_fail(ctx, GD_ERR_SOLVER, "%lld samples outside the byte-index grid: use the u16 path", (long long)nbad);
        }
        return gd_hist2d_prebinned8(ctx, B, d_idx_x, d_idx_y, d_hist);
    }
    const int F = 256;
    for (int c = 0; c < ncols; ++c) GD_REQUIRE(cols[c] >= 0 && cols[c] < ctx->n + GD_EXTRA_COLS && d_idx_out[c], "bad column");
    const BucketRoute R = ncols ? bucket_route(ctx, cols, ncols, binmin, width, d_idx_out) : BucketRoute();
    std::vector<PrebinCol8> hc;
    std::vector<int> fp_of((size_t)ncols, -1);
    for (int c = 0; c < ncols; ++c) {
        if (R.slot_of[c] >= 0) continue;
        PrebinCol8 r;
        r.x = ctx->cols + (int64_t)cols[c] * ctx->ld;
        r.idx = (unsigned char*)d_idx_out[c];
        r.binmin = binmin[c];
        r.width = width[c];
        fp_of[c] = (int)hc.size();
        hc.push_back(r);
    }
    const int nfp = (int)hc.size();
    std::vector<Hist2DPair8> hp((size_t)B);
    for (int b = 0; b < B; ++b) {
        hp[b].ix = (const unsigned char*)d_idx_x[b];
        hp[b].iy = (const unsigned char*)d_idx_y[b];
        GD_REQUIRE(hp[b].ix && hp[b].iy, "null index column");
    }
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_cols = take((int64_t)nfp * sizeof(PrebinCol8)), o_pairs = take((int64_t)B * sizeof(Hist2DPair8));
    const int64_t table_bytes = off;
    const int64_t o_bad = take((int64_t)nfp * 8), o_flags = take((int64_t)B * 4);
    const int64_t zero_end = off;
    const int nchunks = u8_chunks(ctx, B, ctx->N);
    const int64_t o_part = take(nchunks > 1 ? (int64_t)B * nchunks * 32768 * 4 : 0);
    const int64_t o_route = take(bucket_route_bytes(ctx, R));
    char* base = (char*)gd_scratch2(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    {
        std::vector<char> tab((size_t)table_bytes, 0);
        if (nfp) memcpy(tab.data() + o_cols, hc.data(), (size_t)nfp * sizeof(PrebinCol8));
        memcpy(tab.data() + o_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair8));
        GD_TRY(gd_stage_h2d(ctx, base, tab.data(), (size_t)table_bytes));
    }
    GD_HIP(hipMemsetAsync(base + o_bad, 0, (size_t)(zero_end - o_bad), ctx->stream));
    if (nfp) {
        int nblk = (int)((ctx->N / 8 + 255) / 256);
        if (nblk > 2 * ctx->cu_count) nblk = 2 * ctx->cu_count;
        if (nblk < 1) nblk = 1;
        k_prebin8_batch<<<dim3(nblk, nfp), 256, 0, ctx->stream>>>((const PrebinCol8*)(base + o_cols), ctx->N, F,
                                                                 (unsigned long long*)(base + o_bad));
        GD_KERNEL_CHECK();
    }
    unsigned long long* d_bad_bq = nullptr;
    if (!R.recs.empty()) GD_TRY(bucket_route_launch<true>(ctx, R, base + o_route, F, &d_bad_bq));
    const int nblocks = (B * nchunks + 7) / 8 * 8;
    GD_HIP(hipFuncSetAttribute((const void*)k_hist2d_u8_pf<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HIST_BYTES + 128));
    k_hist2d_u8_pf<3><<<nblocks, 1024, LDS_HIST_BYTES + 128, ctx->stream>>>((const Hist2DPair8*)(base + o_pairs), B, ctx->N,
                                                                            (double*)d_hist, (int*)(base + o_flags), nchunks,
                                                                            (unsigned int*)(base + o_part));
    GD_KERNEL_CHECK();
    if (nchunks > 1) {
        k_p8_reduce<<<dim3(32, B), 256, 0, ctx->stream>>>((const unsigned int*)(base + o_part), nchunks, (double*)d_hist);
        GD_KERNEL_CHECK();
    }
    std::vector<unsigned long long> hb((size_t)nfp), hq(R.recs.size());
    std::vector<int> hf((size_t)B);
    if (nfp) GD_TRY(gd_fetch(ctx, hb.data(), base + o_bad, (size_t)nfp * 8));
    if (!R.recs.empty()) GD_TRY(gd_fetch(ctx, hq.data(), d_bad_bq, hq.size() * 8));
    GD_TRY(gd_fetch(ctx, hf.data(), base + o_flags, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    int64_t nbad = 0;
    for (int c = 0; c < ncols; ++c) bad_out[c] = (int64_t)(R.slot_of[c] >= 0 ? hq[R.slot_of[c]] : hb[fp_of[c]]), nbad += bad_out[c];
    if (nbad) return gd_fail(ctx, GD_ERR_SOLVER, "%lld samples outside the byte-index grid: use the u16 path", (long long)nbad);
    for (int b = 0; b < B; ++b)
        if (hf[b]) return gd_fail(ctx, GD_ERR_SOLVER, "16-bit bin counter wrapped in pair %d: redo with gd_hist2d_prebinned", b);
    return GD_OK;
}

int gd_hist2d_sheared(gd_ctx* ctx, int32_t B, const int32_t* coli, const int32_t* colj, const double* r0,
                      const double* r1, const double* xmin, const double* dx, const double* ymin, const double* dy,
                      int32_t F, void* d_hist) {
    GD_REQUIRE(ctx && coli && colj && r0 && r1 && xmin && dx && ymin && dy && d_hist && B > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    std::vector<Hist2DPair> hp((size_t)B);
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(coli[b] >= 0 && coli[b] < ctx->n + GD_EXTRA_COLS && colj[b] >= 0 && colj[b] < ctx->n + GD_EXTRA_COLS, "column out of range");
        Hist2DPair& p = hp[b];
        memset(&p, 0, sizeof p);
        p.x = ctx->cols + (int64_t)coli[b] * ctx->ld;
        p.y = ctx->cols + (int64_t)colj[b] * ctx->ld;
        p.bx = xmin[b], p.wx = dx[b], p.by = ymin[b], p.wy = dy[b], p.r0 = r0[b], p.r1 = r1[b];
    }
    return launch_hist2d<1>(ctx, B, hp, F, (double*)d_hist);
}

int gd_minmax_affine(gd_ctx* ctx, int32_t B, const int32_t* coli, const int32_t* colj, const double* a,
                     const double* b, double* out) {
    GD_REQUIRE(ctx && coli && colj && a && b && out && B > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    std::vector<Hist2DPair> hp((size_t)B);
    for (int q = 0; q < B; ++q) {
        GD_REQUIRE(coli[q] >= 0 && coli[q] < ctx->n + GD_EXTRA_COLS && colj[q] >= 0 && colj[q] < ctx->n + GD_EXTRA_COLS, "column out of range");
        Hist2DPair& p = hp[q];
        memset(&p, 0, sizeof p);
        p.x = ctx->cols + (int64_t)coli[q] * ctx->ld;
        p.y = ctx->cols + (int64_t)colj[q] * ctx->ld;
        p.r0 = a[q], p.r1 = b[q];
    }
    if (B >= 4 && getenv("GDHIP_MINMAX_UNGROUPED") == nullptr) {
        // groups of pairs over shared columns: each column of a group is read once
        std::vector<int> order((size_t)B);
        for (int q = 0; q < B; ++q) order[q] = q;
        std::sort(order.begin(), order.end(), [&](int u, int v) {
            const int ul = std::min(coli[u], colj[u]), uh = std::max(coli[u], colj[u]);
            const int vl = std::min(coli[v], colj[v]), vh = std::max(coli[v], colj[v]);
            return ul != vl ? ul < vl : (uh != vh ? uh < vh : u < v);
        });
        std::vector<MinmaxGroup> groups;
        std::vector<std::vector<int>> members;  // original pair index per slot
        std::vector<int> gcols;                 // column ids of the open group
        auto slot_of = [&](int c) {
            for (size_t k = 0; k < gcols.size(); ++k)
                if (gcols[k] == c) return (int)k;
            return -1;
        };
        for (int q : order) {
            const int ci = coli[q], cj = colj[q];
            bool open = !groups.empty();
            if (open) {
                const int need = (slot_of(ci) < 0) + (cj != ci && slot_of(cj) < 0);
                open = (int)gcols.size() + need <= MMG_COLS && groups.back().npairs < MMG_PAIRS;
            }
            if (!open) {
                MinmaxGroup g;
                memset(&g, 0, sizeof g);
                groups.push_back(g);
                members.emplace_back();
                gcols.clear();
            }
            MinmaxGroup& g = groups.back();
            for (int c : {ci, cj})
                if (slot_of(c) < 0) {
                    g.col[gcols.size()] = ctx->cols + (int64_t)c * ctx->ld;
                    gcols.push_back(c);
                }
            g.ncols = (int)gcols.size();
            g.a[g.npairs] = slot_of(ci), g.b[g.npairs] = slot_of(cj);
            g.r0[g.npairs] = a[q], g.r1[g.npairs] = b[q];
            members.back().push_back(q);
            ++g.npairs;
        }
        const int ng = (int)groups.size();
        int nblk = (6 * ctx->cu_count + ng - 1) / ng;
        if (nblk < 8) nblk = 8;
        if (nblk > 1024) nblk = 1024;
        const int64_t o_part = ((int64_t)ng * sizeof(MinmaxGroup) + 255) / 256 * 256;
        const int64_t part_doubles = (int64_t)ng * nblk * MMG_PAIRS * 2;
        char* base = (char*)gd_scratch(ctx, o_part + part_doubles * 8);
        if (!base) return GD_ERR_NOMEM;
        GD_TRY(gd_h2d(ctx, base, groups.data(), (size_t)ng * sizeof(MinmaxGroup)));
        k_minmax_affine_grouped<<<dim3(nblk, ng), 256, 0, ctx->stream>>>((const MinmaxGroup*)base, ctx->N,
                                                                        (double*)(base + o_part));
        GD_KERNEL_CHECK();
        std::vector<double> h((size_t)part_doubles);
        GD_TRY(gd_fetch(ctx, h.data(), base + o_part, h.size() * 8));
        GD_TRY(gd_stream_sync(ctx));
        for (int g = 0; g < ng; ++g)
            for (int p = 0; p < groups[g].npairs; ++p) {
                double mn = INFINITY, mx = -INFINITY;
                for (int k = 0; k < nblk; ++k) {
                    const double* v = &h[(((size_t)g * nblk + k) * MMG_PAIRS + p) * 2];
                    if (v[0] < mn) mn = v[0];
                    if (v[1] > mx) mx = v[1];
                }
                out[2 * members[g][p]] = mn, out[2 * members[g][p] + 1] = mx;
            }
        return GD_OK;
    }
    int nblk = (4 * ctx->cu_count + B - 1) / B;
    if (nblk < 8) nblk = 8;
    if (nblk > 1024) nblk = 1024;
    const int64_t o_part = ((int64_t)B * sizeof(Hist2DPair) + 255) / 256 * 256;
    char* base = (char*)gd_scratch(ctx, o_part + (int64_t)B * nblk * 16);
    if (!base) return GD_ERR_NOMEM;
    Hist2DPair* d_pairs = (Hist2DPair*)base;
    double* d_part = (double*)(base + o_part);
    GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(Hist2DPair)));
    k_minmax_affine<<<dim3(nblk, B), 256, 0, ctx->stream>>>(d_pairs, ctx->N, d_part);
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)B * nblk * 2);
    GD_TRY(gd_fetch(ctx, h.data(), d_part, h.size() * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int q = 0; q < B; ++q) {
        double mn = INFINITY, mx = -INFINITY;
        for (int k = 0; k < nblk; ++k) {
            const double v0 = h[((size_t)q * nblk + k) * 2], v1 = h[((size_t)q * nblk + k) * 2 + 1];
            if (v0 < mn) mn = v0;
            if (v1 > mx) mx = v1;
        }
        out[2 * q] = mn, out[2 * q + 1] = mx;
    }
    return GD_OK;
}

}  // extern "C"
