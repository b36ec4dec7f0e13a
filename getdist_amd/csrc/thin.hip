// Weight-one thinning of integer-weight chains and the statistics of the thinned chains that the Raftery-Lewis and
// CorrSteps convergence tests need (chains.py:878-916; mcsamples.py:1039-1221).
//
// The reference materialises thin_ix with a Python loop / np.unique.  With C_i the inclusive cumulative weight,
// row i is emitted  C_i/f - C_{i-1}/f  times (or, when f >= max weight, row 0 plus every row where C_i/f steps), and
// its first output position is  C_{i-1}/f  -- so ONE prefix sum of the weights (cached per sample set) turns every
// later thinning, for any factor and any chain [lo,hi), into an embarrassingly parallel scatter.
#include "ctx.hpp"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ long long w_int(const double* __restrict__ w, int64_t i) { return w ? (long long)w[i] : 1LL; }

__global__ void __launch_bounds__(SCAN_THREADS) k_wscan_reduce(const double* __restrict__ w, int64_t N,
                                                                long long* __restrict__ bsum) {
    __shared__ long long red[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    long long s = 0;
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        const int64_t i = base + (int64_t)q * SCAN_THREADS + threadIdx.x;
        if (i < N) s += w_int(w, i);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int i = 0; i < SCAN_THREADS / 64; ++i) t += red[i];
        bsum[blockIdx.x] = t;
    }
}

// exclusive scan of the tile sums, one block
__global__ void __launch_bounds__(1024) k_wscan_blocks(long long* __restrict__ bsum, int nb) {
    __shared__ long long sh[1024];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nb; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const long long v = (i < nb) ? bsum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long long a = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nb) bsum[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
}

// inclusive cumulative weights: each thread owns SCAN_ITEMS consecutive rows of its tile
__global__ void __launch_bounds__(SCAN_THREADS) k_wscan_down(const double* __restrict__ w, int64_t N,
                                                              const long long* __restrict__ boff,
                                                              long long* __restrict__ C) {
    __shared__ long long sh[SCAN_THREADS];
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    long long v[SCAN_ITEMS];
    long long s = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        const int64_t i = i0 + q;
        s += (i < N) ? w_int(w, i) : 0;
        v[q] = s;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {
        const long long a = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += a;
        __syncthreads();
    }
    const long long before = boff[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q)
        if (i0 + q < N) C[i0 + q] = before + v[q];
}

// scatter the thinned row list of chain [lo,hi)
__global__ void k_thin_rows(const long long* __restrict__ C, int64_t lo, int64_t hi, long long f, int unique_mode,
                            int32_t* __restrict__ rows) {
    const long long base = lo > 0 ? C[lo - 1] : 0;
    const long long v0 = (C[lo] - base) / f;
    for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
        const long long v = (C[i] - base) / f;
        const long long vp = (i > lo) ? (C[i - 1] - base) / f : 0;
        if (unique_mode) {  // np.unique(cumsum // f, return_index=True): first row of every distinct value
            if (i == lo)
                rows[0] = (int32_t)i;
            else if (v > vp)
                rows[1 + vp - v0] = (int32_t)i;
        } else {
            for (long long q = vp; q < v; ++q) rows[q] = (int32_t)i;
        }
    }
}

// For every column and threshold: counts of (b[k-2], b[k-1], b[k]) triples (8) and (b[k-1], b[k]) pairs (4) of the
// binary chain b[k] = (x[rows[k]] >= u) ? 0 : 1  (mcsamples.py:1063-1066, 1120-1122).  grid (blocks, ncols)
#define MAX_THR 4
__global__ void __launch_bounds__(256) k_binary_transitions(const double* __restrict__ cols, int64_t ld,
                                                            const int32_t* __restrict__ colidx,
                                                            const int32_t* __restrict__ rows, int64_t K,
                                                            const double* __restrict__ thr, int nthr,
                                                            unsigned long long* __restrict__ counts) {
    const double* x = cols + (int64_t)colidx[blockIdx.y] * ld;
    double u[MAX_THR];
    for (int t = 0; t < MAX_THR; ++t) u[t] = (t < nthr) ? thr[(int64_t)blockIdx.y * nthr + t] : 0.0;
    unsigned int cnt[MAX_THR][12];
    for (int t = 0; t < MAX_THR; ++t)
        for (int c = 0; c < 12; ++c) cnt[t][c] = 0;
    for (int64_t k = 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (int64_t)gridDim.x * blockDim.x) {
        const double x0 = x[rows[k]], x1 = x[rows[k - 1]];
        const double x2 = (k >= 2) ? x[rows[k - 2]] : 0.0;
#pragma unroll
        for (int t = 0; t < MAX_THR; ++t) {
            if (t >= nthr) break;
            const int b0 = (x0 >= u[t]) ? 0 : 1, b1 = (x1 >= u[t]) ? 0 : 1;
            const int pair = b1 * 2 + b0;
#pragma unroll
            for (int c = 0; c < 4; ++c) cnt[t][8 + c] += (pair == c);
            if (k >= 2) {
                const int tri = ((x2 >= u[t]) ? 0 : 4) + pair;
#pragma unroll
                for (int c = 0; c < 8; ++c) cnt[t][c] += (tri == c);
            }
        }
    }
    for (int t = 0; t < nthr; ++t)
        for (int c = 0; c < 12; ++c) {
            unsigned int v = cnt[t][c];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[((int64_t)blockIdx.y * nthr + t) * 12 + c], (unsigned long long)v);
        }
}

// part[(c*maxoff + off-1)*nblk + b] = partial of sum_k (x[rows[k+off]]-m)(x[rows[k]]-m).  grid (nblk, maxoff, ncols)
__global__ void __launch_bounds__(256) k_thinned_lag(const double* __restrict__ cols, int64_t ld,
                                                     const int32_t* __restrict__ colidx, const double* __restrict__ means,
                                                     const int32_t* __restrict__ rows, int64_t K,
                                                     double* __restrict__ part) {
    __shared__ double red[4];
    const int off = blockIdx.y + 1;
    const double* x = cols + (int64_t)colidx[blockIdx.z] * ld;
    const double m = means[blockIdx.z];
    double s = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k + off < K; k += (int64_t)gridDim.x * blockDim.x)
        s += (x[rows[k + off]] - m) * (x[rows[k]] - m);
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

static int ensure_weight_cumsum(gd_ctx* ctx) {
    if (ctx->wcum) return GD_OK;
    const int64_t N = ctx->N;
    const int nb = (int)((N + SCAN_TILE - 1) / SCAN_TILE);
    long long* C = nullptr;
    GD_HIP(hipMalloc((void**)&C, (size_t)(N * 8)));
    long long* bsum = (long long*)gd_scratch(ctx, (int64_t)nb * 8);
    if (!bsum) {
        (void)hipFree(C);
        return GD_ERR_NOMEM;
    }
    const double* w = ctx->w_sel ? ctx->w_main : ctx->w;  // always the sample weights
    k_wscan_reduce<<<nb, SCAN_THREADS, 0, ctx->stream>>>(w, N, bsum);
    k_wscan_blocks<<<1, 1024, 0, ctx->stream>>>(bsum, nb);
    k_wscan_down<<<nb, SCAN_THREADS, 0, ctx->stream>>>(w, N, bsum, C);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(C);
        return gd_fail(ctx, GD_ERR_HIP, "cumulative-weight scan failed");
    }
    ctx->wcum = C;
    return GD_OK;
}

extern "C" {

int gd_weights_integral(gd_ctx* ctx, int32_t* out) {
    GD_REQUIRE(ctx && out, "null argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    const bool has_w = (ctx->w_sel ? ctx->w_main : ctx->w) != nullptr;
    *out = (!has_w || (ctx->w_sel ? ctx->w_main_integral : ctx->w_integral)) ? 1 : 0;
    return GD_OK;
}

int gd_thin_rows(gd_ctx* ctx, int64_t lo, int64_t hi, int64_t factor, int32_t unique_mode, void* d_rows, int64_t capacity,
                 int64_t* count_out) {
    GD_REQUIRE(ctx && d_rows && count_out, "null argument");
    GD_REQUIRE(ctx->cols && lo >= 0 && hi <= ctx->N && lo < hi, "bad row range");
    GD_REQUIRE(factor >= 1, "Thin factor must be a positive integer");
    GD_REQUIRE(ctx->N < 2147483647LL, "row indices are 32-bit");
    int32_t integral = 0;
    gd_weights_integral(ctx, &integral);
    GD_REQUIRE(integral, "Can only thin with integer weights");
    int rc = ensure_weight_cumsum(ctx);
    if (rc) return rc;
    long long ends[3] = {0, 0, 0};  // C[lo-1], C[lo], C[hi-1]
    if (lo > 0) GD_TRY(gd_fetch(ctx, &ends[0], ctx->wcum + lo - 1, 8));
    GD_TRY(gd_fetch(ctx, &ends[1], ctx->wcum + lo, 8));
    GD_TRY(gd_fetch(ctx, &ends[2], ctx->wcum + hi - 1, 8));
    GD_TRY(gd_stream_sync(ctx));
    const long long vlast = (ends[2] - ends[0]) / factor, v0 = (ends[1] - ends[0]) / factor;
    const int64_t K = unique_mode ? 1 + vlast - v0 : vlast;
    *count_out = K;
    GD_REQUIRE(K <= capacity, "thinned row buffer too small");
    if (K == 0) return GD_OK;
    k_thin_rows<<<2048, 256, 0, ctx->stream>>>(ctx->wcum, lo, hi, factor, unique_mode, (int32_t*)d_rows);
    GD_KERNEL_CHECK();
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_binary_transitions(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const void* d_rows, int64_t K,
                          const double* thresholds, int32_t nthr, int64_t* counts_out) {
    GD_REQUIRE(ctx && cols && d_rows && thresholds && counts_out && ncols > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(nthr >= 1 && nthr <= MAX_THR, "1..4 thresholds per column");
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    const int64_t nc = (int64_t)ncols * nthr * 12;
    memset(counts_out, 0, (size_t)nc * 8);
    if (K < 2) return GD_OK;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_idx = take((int64_t)ncols * 4), o_thr = take((int64_t)ncols * nthr * 8), o_cnt = take(nc * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    GD_TRY(gd_h2d(ctx, base + o_idx, cols, (size_t)ncols * 4));
    GD_TRY(gd_h2d(ctx, base + o_thr, thresholds, (size_t)ncols * nthr * 8));
    GD_HIP(hipMemsetAsync(base + o_cnt, 0, (size_t)nc * 8, ctx->stream));
    int nblk = (int)((K + 255) / 256);
    if (nblk > 512) nblk = 512;
    k_binary_transitions<<<dim3(nblk, ncols), 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, (const int32_t*)(base + o_idx),
                                                                    (const int32_t*)d_rows, K, (const double*)(base + o_thr),
                                                                    nthr, (unsigned long long*)(base + o_cnt));
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, counts_out, base + o_cnt, (size_t)nc * 8));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

int gd_thinned_lag_sums(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* means, const void* d_rows, int64_t K,
                        int32_t maxoff, double* out) {
    GD_REQUIRE(ctx && cols && means && d_rows && out && ncols > 0 && maxoff > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(maxoff < 65536 && ncols < 65536, "too many lags / columns");
    for (int i = 0; i < ncols; ++i) GD_REQUIRE(cols[i] >= 0 && cols[i] < ctx->n + GD_EXTRA_COLS, "column out of range");
    const int nblk = 64;
    const int64_t np = (int64_t)ncols * maxoff * nblk;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_idx = take((int64_t)ncols * 4), o_mean = take((int64_t)ncols * 8), o_part = take(np * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    GD_TRY(gd_h2d(ctx, base + o_idx, cols, (size_t)ncols * 4));
    GD_TRY(gd_h2d(ctx, base + o_mean, means, (size_t)ncols * 8));
    k_thinned_lag<<<dim3(nblk, maxoff, ncols), 256, 0, ctx->stream>>>(ctx->cols, ctx->ld, (const int32_t*)(base + o_idx),
                                                                     (const double*)(base + o_mean), (const int32_t*)d_rows, K,
                                                                     (double*)(base + o_part));
    GD_KERNEL_CHECK();
    std::vector<double> h((size_t)np);
    GD_TRY(gd_fetch(ctx, h.data(), base + o_part, (size_t)np * 8));
    GD_TRY(gd_stream_sync(ctx));
    for (int64_t e = 0; e < (int64_t)ncols * maxoff; ++e) {
        double s = 0;
        for (int b = 0; b < nblk; ++b) s += h[(size_t)(e * nblk + b)];
        out[e] = s;
    }
    return GD_OK;
}

}  // extern "C"
