// 2D density assembly on the device for a batch of pairs sharing the fine grid size F
// (mcsamples.py:1857-1990): Gaussian window synthesis, zero-padded linear convolution, linear boundary correction,
// multiplicative bias correction, max-normalisation.  The convolutions run through hand-written transforms in LDS
// (k_rows_fwd / k_win_spec / k_col_conv / k_rows_inv below); frames above 512, explicit prior masks and periodic axes
// take the older route through rocFFT frames (fft.hip).
//
// Frame convention (both routes): every operand lives in an S x S real frame (S >= F + 2*winw on a ladder of
// transform-friendly sizes).  The (F+2w)^2 prior mask sits at the frame origin, the F^2 histogram at offset (w,w), and
// each window is stored centred-with-wrap, so that one circular convolution gives the reference's 'same' result for
// the histogram and its 'valid' result for the mask at frame positions [w, F+w)^2 (convolve.py:405-444).
#include "ctx.hpp"

struct D2Pair {
    double c00, c11, c10;  // inverse bandwidth matrix (mcsamples.py:1864)
    int w;                 // winw
    int flags;             // bit0/1 x bot/top, bit2/3 y bot/top, bit4/5 x/y periodic, bit6 boundary correction applies
    int hidx;              // which histogram of the source block this pair convolves (the batch need not be contiguous)
    int pad;
};

__device__ __forceinline__ double win_raw(const D2Pair& p, int i1, int i2) {
    // mcsamples.py:1865-1866: i1 = row (y) offset, i2 = column (x) offset
    const double a = (double)(i1 * i1) * p.c00 + (double)(i2 * i2) * p.c11 + 2.0 * p.c10 * (double)(i1 * i2);
    return exp(-a / 2.0);
}

__global__ void k_win_sum(const D2Pair* __restrict__ pairs, double* __restrict__ wsum) {
    __shared__ double red[16];
    const D2Pair p = pairs[blockIdx.x];
    const int M = 2 * p.w + 1;
    double s = 0;
    for (int e = threadIdx.x; e < M * M; e += blockDim.x) s += win_raw(p, e / M - p.w, e % M - p.w);
    s = block_sum(s, red);
    if (threadIdx.x == 0) wsum[blockIdx.x] = s;
}

// window * x^px * y^py, centred with wrap, in an S x S frame (zero elsewhere). grid (blocks, B)
__global__ void k_fill_window(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, int S, int px, int py,
                              double* __restrict__ frames) {
    const D2Pair p = pairs[blockIdx.y];
    double* fr = frames + (int64_t)blockIdx.y * S * S;
    const double ws = wsum[blockIdx.y];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S * S; e += gridDim.x * blockDim.x) {
        const int r = e / S, c = e % S;
        const int i1 = (r <= S / 2) ? r : r - S, i2 = (c <= S / 2) ? c : c - S;
        double v = 0;
        if (i1 >= -p.w && i1 <= p.w && i2 >= -p.w && i2 <= p.w) {
            v = win_raw(p, i1, i2) / ws;
            for (int q = 0; q < px; ++q) v = v * (double)i2;
            for (int q = 0; q < py; ++q) v = v * (double)i1;
        }
        fr[e] = v;
    }
}

// histogram (or any F x F array) placed at offset (w,w)
__global__ void k_fill_embed(const D2Pair* __restrict__ pairs, const double* __restrict__ src, int F, int S,
                             double* __restrict__ frames) {
    const int w = pairs[blockIdx.y].w;
    const double* s = src + (int64_t)pairs[blockIdx.y].hidx * F * F;
    double* fr = frames + (int64_t)blockIdx.y * S * S;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S * S; e += gridDim.x * blockDim.x) {
        const int r = e / S - w, c = e % S - w;
        fr[e] = (r >= 0 && r < F && c >= 0 && c < F) ? s[(int64_t)r * F + c] : 0.0;
    }
}

// prior masks (mcsamples.py:1688-1712) at the frame origin, size (F+2w)^2, zero outside.
// kind 0: edge mask of _setEdgeMask2D;  kind 1: the same followed by _setAllEdgeMask2D (edge_applied says
// whether the half-weights of kind 0 were applied before).
__device__ __forceinline__ double mask_1d(int p, int F, int w, bool bot, bool top, bool all_edges) {
    if (p < 0 || p >= F + 2 * w) return 0.0;
    double v = 1.0;
    if (bot) {
        if (p == w) v = 0.5;
        if (p < w) v = 0.0;
    }
    if (top) {
        if (p == F + w - 1) v = 0.5;
        if (p > F + w - 1) v = 0.0;
    }
    if (all_edges && (p < w || p >= F + w)) v = 0.0;
    return v;
}
__global__ void k_fill_mask(const D2Pair* __restrict__ pairs, int F, int S, int kind, int edge_applied,
                            double* __restrict__ frames) {
    const D2Pair p = pairs[blockIdx.y];
    double* fr = frames + (int64_t)blockIdx.y * S * S;
    const bool use_edges = (kind == 0) || edge_applied;
    const bool xb = use_edges && (p.flags & 1), xt = use_edges && (p.flags & 2), yb = use_edges && (p.flags & 4),
               yt = use_edges && (p.flags & 8);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S * S; e += gridDim.x * blockDim.x) {
        const int r = e / S, c = e % S;
        fr[e] = mask_1d(c, F, p.w, xb, xt, kind == 1 && !(p.flags & 16)) *
                mask_1d(r, F, p.w, yb, yt, kind == 1 && !(p.flags & 32));
    }
}

// ---- prior-mask moments without FFTs -------------------------------------------------------------------------
// The prior masks of mcsamples.py:1688-1712 are axis-aligned boxes (1 inside, 1/2 on a limit edge, 0 beyond), so
// a_pq = conv(mask, Win * x^p y^q) at a pixel is a rectangle sum of the window moment: four look-ups in its
// summed-area table, minus half of the edge row / column, plus a quarter of the corner cell.  This replaces the
// mask FFT and one inverse FFT per moment (mcsamples.py:1905-1919, 1964-1966) with (2w+2)^2 table entries per pair.
struct MomSpec {
    int px, py, kind;  // kind 0: edge mask, kind 1: all-edge mask
    double* dst;       // B x F x F
};
struct MomList {
    int n, edge_applied;
    MomSpec m[7];
};

// grid (n moments, B), block 256: sat[(a+1)*(M+1) + (b+1)] = sum_{a'<=a, b'<=b} K[a'-w][b'-w]
__global__ void k_window_sat(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, MomList L,
                             int64_t sat_stride, double* __restrict__ sat) {
    const D2Pair p = pairs[blockIdx.y];
    const MomSpec sp = L.m[blockIdx.x];
    if (sp.kind == 0 && !(p.flags & 64)) return;
    const int w = p.w, M = 2 * w + 1, M1 = M + 1;
    const double ws = wsum[blockIdx.y];
    double* s = sat + ((int64_t)blockIdx.y * L.n + blockIdx.x) * sat_stride;
    for (int t = threadIdx.x; t < M1; t += blockDim.x) s[t] = 0.0, s[(int64_t)t * M1] = 0.0;
    // column prefixes (thread per column: coalesced)
    for (int c = threadIdx.x; c < M; c += blockDim.x) {
        const int i2 = c - w;
        double run = 0;
        for (int r = 0; r < M; ++r) {
            const int i1 = r - w;
            double v = win_raw(p, i1, i2) / ws;
            for (int q = 0; q < sp.px; ++q) v = v * (double)i2;
            for (int q = 0; q < sp.py; ++q) v = v * (double)i1;
            run += v;
            s[(int64_t)(r + 1) * M1 + c + 1] = run;
        }
    }
    __syncthreads();
    // row prefixes
    for (int r = threadIdx.x; r < M; r += blockDim.x) {
        double* row = s + (int64_t)(r + 1) * M1 + 1;
        double run = 0;
        for (int c = 0; c < M; ++c) {
            run += row[c];
            row[c] = run;
        }
    }
}

__device__ __forceinline__ double sat_rect(const double* __restrict__ s, int M1, int a0, int a1, int b0, int b1) {
    return ((s[(a1 + 1) * M1 + b1 + 1] - s[a0 * M1 + b1 + 1]) - s[(a1 + 1) * M1 + b0]) + s[a0 * M1 + b0];
}

// one axis of a mask: support [lo, hi] in padded coordinates, half weight on lo / hi when a limit sits there
struct MaskIv {
    int lo, hi;
    bool hlo, hhi;
};
__device__ __forceinline__ MaskIv mask_interval(int F, int w, bool bot, bool top, int kind, bool use_edges) {
    MaskIv iv;
    iv.lo = (kind == 1 || (use_edges && bot)) ? w : 0;
    iv.hi = (kind == 1 || (use_edges && top)) ? F + w - 1 : F + 2 * w - 1;
    iv.hlo = use_edges && bot;
    iv.hhi = use_edges && top;
    return iv;
}

// Which entries of a moment's summed-area table a pixel's mask moment is made of (the same for every moment of one mask).
struct MaskGeom {
    int a0, a1, b0, b1, ner, nec, er[2], ec[2];
    bool any, full;  // any: the window meets the mask at all; full: the whole window lies inside it, away from any half row
};
__device__ __forceinline__ MaskGeom mask_geom(int F, int w, const MaskIv& ix, const MaskIv& iy, int x, int y) {
    MaskGeom g;
    // window offset i covers padded position  (pixel + w - i)
    const int i_lo = max(-w, y + w - iy.hi), i_hi = min(w, y + w - iy.lo);
    const int j_lo = max(-w, x + w - ix.hi), j_hi = min(w, x + w - ix.lo);
    g.any = i_lo <= i_hi && j_lo <= j_hi;
    g.ner = g.nec = 0;
    g.a0 = i_lo + w, g.a1 = i_hi + w, g.b0 = j_lo + w, g.b1 = j_hi + w;
    if (g.any) {
        if (iy.hlo && y + w - iy.lo <= w) g.er[g.ner++] = y + 2 * w - iy.lo;
        if (iy.hhi && y + w - iy.hi >= -w) g.er[g.ner++] = y + 2 * w - iy.hi;
        if (ix.hlo && x + w - ix.lo <= w) g.ec[g.nec++] = x + 2 * w - ix.lo;
        if (ix.hhi && x + w - ix.hi >= -w) g.ec[g.nec++] = x + 2 * w - ix.hi;
    }
    g.full = g.any && g.ner == 0 && g.nec == 0 && g.a0 == 0 && g.b0 == 0 && g.a1 == 2 * w && g.b1 == 2 * w;
    return g;
}
// the moment itself from its table: rectangle sum, minus half of the edge rows / columns, plus a quarter of the corners
__device__ __forceinline__ double mask_moment(const double* __restrict__ s, int M1, const MaskGeom& g) {
    if (!g.any) return 0.0;
    double v = sat_rect(s, M1, g.a0, g.a1, g.b0, g.b1);
    for (int k = 0; k < g.ner; ++k) v -= 0.5 * sat_rect(s, M1, g.er[k], g.er[k], g.b0, g.b1);
    for (int k = 0; k < g.nec; ++k) v -= 0.5 * sat_rect(s, M1, g.a0, g.a1, g.ec[k], g.ec[k]);
    for (int k = 0; k < g.ner; ++k)
        for (int l = 0; l < g.nec; ++l) v += 0.25 * sat_rect(s, M1, g.er[k], g.er[k], g.ec[l], g.ec[l]);
    return v;
}

// ---- class tables of the mask moments (round 5) -----------------------------------------------------------------
// mask_geom(x, y) depends on x only through how the window is clipped by the mask's x interval: not at all for
// f_lo <= x <= f_hi (the `full` columns), and otherwise through the distance to the clipping edge -- at most w + 1 columns
// on each side.  So a moment takes (2w + 3)^2 distinct values over the grid: class c = x for x < f_lo, w + 1 for the full
// columns, w + 1 + (x - f_hi) beyond (likewise in y).  k_mask_tables evaluates mask_moment once per class pair from the
// summed-area table -- the same function on the same operands as the per-pixel evaluation, hence the same bits -- and the
// consumers (k_boundary<true>, k_rows_inv<1>) look a pixel's moments up instead of running the geometry and up to nine
// rectangle sums per moment on every border pixel (and staging 9-55 KB of summed-area tables per block to do so).
// Needs f_lo <= f_hi + 1 on both axes (no pixel clipped on both sides): the host uses the tables for F >= 4 w + 8.
__device__ __forceinline__ int mask_class(int x, int f_lo, int f_hi, int w) {
    return x < f_lo ? x : (x > f_hi ? w + 1 + (x - f_hi) : w + 1);
}
__device__ __forceinline__ int mask_class_rep(int c, int f_lo, int f_hi, int w) {  // a pixel of class c
    return c <= w ? c : (c == w + 1 ? f_lo : f_hi + (c - w - 1));
}
// grid (n moments, B), after k_window_sat: tab[(b * n + q) * tab_stride + cy * (2w + 3) + cx]
__global__ void __launch_bounds__(256) k_mask_tables(const D2Pair* __restrict__ pairs, MomList L, int F, int64_t sat_stride,
                                                     const double* __restrict__ sat, int64_t tab_stride, double* __restrict__ tab) {
    const D2Pair p = pairs[blockIdx.y];
    const MomSpec sp = L.m[blockIdx.x];
    if (sp.kind == 0 && !(p.flags & 64)) return;
    const int w = p.w, M1 = 2 * w + 2, NC = 2 * w + 3;
    const double* s = sat + ((int64_t)blockIdx.y * L.n + blockIdx.x) * sat_stride;
    double* T = tab + ((int64_t)blockIdx.y * L.n + blockIdx.x) * tab_stride;
    const bool use_edges = (sp.kind == 0) || L.edge_applied;
    const MaskIv ix = mask_interval(F, w, p.flags & 1, p.flags & 2, sp.kind, use_edges);
    const MaskIv iy = mask_interval(F, w, p.flags & 4, p.flags & 8, sp.kind, use_edges);
    const int xf_lo = ix.lo + (ix.hlo ? 1 : 0), xf_hi = ix.hi - 2 * w - (ix.hhi ? 1 : 0);
    const int yf_lo = iy.lo + (iy.hlo ? 1 : 0), yf_hi = iy.hi - 2 * w - (iy.hhi ? 1 : 0);
    for (int e = threadIdx.x; e < NC * NC; e += blockDim.x) {
        const int cy = e / NC, cx = e - cy * NC;
        const int x = mask_class_rep(cx, xf_lo, xf_hi, w), y = mask_class_rep(cy, yf_lo, yf_hi, w);
        const bool valid = x >= 0 && x < F && y >= 0 && y < F && mask_class(x, xf_lo, xf_hi, w) == cx && mask_class(y, yf_lo, yf_hi, w) == cy;
        T[e] = valid ? mask_moment(s, M1, mask_geom(F, w, ix, iy, x, y)) : 0.0;
    }
}

// k_window_sat + k_mask_tables in one kernel with the summed-area table in LDS (dynamic: (2w+2)^2 doubles of the widest
// window of the batch): the table kernel alone took 27 us per 136-pair batch reading the tables back from L2 entry by
// entry -- as long as what the class tables saved in their consumers.  Same operations in the same order as the two
// kernels (column prefixes, row prefixes, mask_moment per class pair), hence the same bits; the summed-area table is still
// written out for the consumers' fallback paths.  grid (n moments, B)
__global__ void __launch_bounds__(256) k_window_sat_tab(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, MomList L,
                                                        int F, int64_t sat_stride, double* __restrict__ sat, int64_t tab_stride,
                                                        double* __restrict__ tab) {
    extern __shared__ double sl[];
    const D2Pair p = pairs[blockIdx.y];
    const MomSpec sp = L.m[blockIdx.x];
    if (sp.kind == 0 && !(p.flags & 64)) return;
    const int w = p.w, M = 2 * w + 1, M1 = M + 1, NC = 2 * w + 3;
    const double ws = wsum[blockIdx.y];
    for (int t = threadIdx.x; t < M1; t += blockDim.x) sl[t] = 0.0, sl[t * M1] = 0.0;
    for (int c = threadIdx.x; c < M; c += blockDim.x) {
        const int i2 = c - w;
        double run = 0;
        for (int r = 0; r < M; ++r) {
            const int i1 = r - w;
            double v = win_raw(p, i1, i2) / ws;
            for (int q = 0; q < sp.px; ++q) v = v * (double)i2;
            for (int q = 0; q < sp.py; ++q) v = v * (double)i1;
            run += v;
            sl[(r + 1) * M1 + c + 1] = run;
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < M; r += blockDim.x) {
        double* row = sl + (r + 1) * M1 + 1;
        double run = 0;
        for (int c = 0; c < M; ++c) {
            run += row[c];
            row[c] = run;
        }
    }
    __syncthreads();
    double* sg = sat + ((int64_t)blockIdx.y * L.n + blockIdx.x) * sat_stride;
    for (int e = threadIdx.x; e < M1 * M1; e += blockDim.x) sg[e] = sl[e];
    double* T = tab + ((int64_t)blockIdx.y * L.n + blockIdx.x) * tab_stride;
    const bool use_edges = (sp.kind == 0) || L.edge_applied;
    const MaskIv ix = mask_interval(F, w, p.flags & 1, p.flags & 2, sp.kind, use_edges);
    const MaskIv iy = mask_interval(F, w, p.flags & 4, p.flags & 8, sp.kind, use_edges);
    const int xf_lo = ix.lo + (ix.hlo ? 1 : 0), xf_hi = ix.hi - 2 * w - (ix.hhi ? 1 : 0);
    const int yf_lo = iy.lo + (iy.hlo ? 1 : 0), yf_hi = iy.hi - 2 * w - (iy.hhi ? 1 : 0);
    for (int e = threadIdx.x; e < NC * NC; e += blockDim.x) {
        const int cy = e / NC, cx = e - cy * NC;
        const int x = mask_class_rep(cx, xf_lo, xf_hi, w), y = mask_class_rep(cy, yf_lo, yf_hi, w);
        const bool valid = x >= 0 && x < F && y >= 0 && y < F && mask_class(x, xf_lo, xf_hi, w) == cx && mask_class(y, yf_lo, yf_hi, w) == cy;
        T[e] = valid ? mask_moment(sl, M1, mask_geom(F, w, ix, iy, x, y)) : 0.0;
    }
}

// grid (blocks, n moments, B): the moments as F x F arrays (the rocFFT route and explicit masks read them; the LDS route
// evaluates them where they are used: k_boundary<true>, k_rows_inv<1>)
__global__ void k_mask_eval(const D2Pair* __restrict__ pairs, MomList L, int F, int64_t sat_stride,
                            const double* __restrict__ sat) {
    const D2Pair p = pairs[blockIdx.z];
    const MomSpec sp = L.m[blockIdx.y];
    if (sp.kind == 0 && !(p.flags & 64)) return;
    const int w = p.w, M1 = 2 * w + 2;
    const double* s = sat + ((int64_t)blockIdx.z * L.n + blockIdx.y) * sat_stride;
    double* dst = sp.dst + (int64_t)blockIdx.z * F * F;
    const bool use_edges = (sp.kind == 0) || L.edge_applied;
    const MaskIv ix = mask_interval(F, w, p.flags & 1, p.flags & 2, sp.kind, use_edges);
    const MaskIv iy = mask_interval(F, w, p.flags & 4, p.flags & 8, sp.kind, use_edges);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * F; e += gridDim.x * blockDim.x) {
        const int y = e / F, x = e % F;
        dst[e] = mask_moment(s, M1, mask_geom(F, w, ix, iy, x, y));
    }
}

// out = a * b * scale (complex), n elements per batch entry
__global__ void k_cmul(const double2* __restrict__ a, const double2* __restrict__ b, int64_t n, double scale,
                       double2* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 x = a[i], y = b[i];
        out[i] = make_double2((x.x * y.x - x.y * y.y) * scale, (x.x * y.y + x.y * y.x) * scale);
    }
}

// per-grid maximum in PM_PARTS partial maxima, one per block of the (PM_PARTS, B) grids the F x F passes run on: the
// passes that produce a grid (crop, bias-correction update, boundary correction) leave their block maxima behind, so no
// separate reduction pass reads the grid again; consumers fold the parts with pair_max()
#define PM_PARTS 64
// crop frame positions [w, F+w)^2 into an F x F array; MODE 1: as the multiplicative bias-correction update
// dst = dst * crop / a00 (mcsamples.py:1972-1973) without materialising the cropped convolution; mx != nullptr: the
// block's maximum of what it wrote goes to mx[b * PM_PARTS + blockIdx.x] (grid (PM_PARTS, B))
template <int MODE>
__global__ void __launch_bounds__(256) k_crop_fused(const D2Pair* __restrict__ pairs, const double* __restrict__ frames, int F, int S,
                                                    double* __restrict__ dst, const double* __restrict__ a00,
                                                    double* __restrict__ mx) {
    __shared__ double red[16];
    const int w = pairs[blockIdx.y].w;
    const double* fr = frames + (int64_t)blockIdx.y * S * S;
    const int64_t o = (int64_t)blockIdx.y * F * F;
    double* d = dst + o;
    double m = -INFINITY;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * F; e += gridDim.x * blockDim.x) {
        double v = fr[(int64_t)(e / F + w) * S + (e % F + w)];
        if (MODE == 1) v = (d[e] * v) / a00[o + e];
        d[e] = v;
        m = fmax(m, v);
    }
    if (mx) {
        m = block_max(m, red);
        if (threadIdx.x == 0) mx[(int64_t)blockIdx.y * PM_PARTS + blockIdx.x] = m;
    }
}

// crop frame positions [w, F+w)^2 into an F x F array
__global__ void k_crop(const D2Pair* __restrict__ pairs, const double* __restrict__ frames, int F, int S,
                       double* __restrict__ dst) {
    const int w = pairs[blockIdx.y].w;
    const double* fr = frames + (int64_t)blockIdx.y * S * S;
    double* d = dst + (int64_t)blockIdx.y * F * F;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * F; e += gridDim.x * blockDim.x)
        d[e] = fr[(int64_t)(e / F + w) * S + (e % F + w)];
}

__global__ void k_pair_max(const double* __restrict__ a, int FF, double* __restrict__ mx) {
    __shared__ double red[16];
    const double* p = a + (int64_t)blockIdx.y * FF;
    const int per = (FF + PM_PARTS - 1) / PM_PARTS, lo = blockIdx.x * per, hi = min(FF, lo + per);
    double m = -INFINITY;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) m = fmax(m, p[i]);
    m = block_max(m, red);
    if (threadIdx.x == 0) mx[(int64_t)blockIdx.y * PM_PARTS + blockIdx.x] = m;
}
__device__ __forceinline__ double pair_max(const double* __restrict__ mx, int b) {
    double m = mx[(int64_t)b * PM_PARTS];
#pragma unroll
    for (int k = 1; k < PM_PARTS; ++k) m = fmax(m, mx[(int64_t)b * PM_PARTS + k]);
    return m;
}

struct BcArrays {
    double *P, *a00, *a10, *a01, *a20, *a02, *a11, *xP, *yP;
};

// linear boundary correction (mcsamples.py:1921-1961), in place on P; only pairs with a limit
// mx_out (may be nullptr; must not alias mx: other blocks of the pair are still folding mx): the block maxima of the
// grid after the correction (grid (PM_PARTS, B)).
// FUSED: the prior-mask moments a_pq are not read from F x F arrays but evaluated here from the pair's summed-area tables
// (sat: moment q of pair b at (b * nmom + q) * sat_stride; the six moments share one MaskGeom, interior pixels -- the whole
// window inside the mask -- take the tables' totals from registers, and the higher moments are only formed where the pixel
// passes the threshold): the same operations in the same order as k_mask_eval + the array form, without writing and
// re-reading six grids per bounded pair.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_boundary(const D2Pair* __restrict__ pairs, BcArrays A, const double* __restrict__ mx, int FF,
                                                  int bco, double* __restrict__ mx_out, const double* __restrict__ sat,
                                                  int64_t sat_stride, int nmom, int F, int stage_lds = 0,
                                                  const double* __restrict__ tab = nullptr, int64_t tab_stride = 0) {
    __shared__ double red[16];
    const int b = blockIdx.y;
    const D2Pair p = pairs[b];
    if ((p.flags & 64) == 0) {  // untouched grid: its maxima move to the output set unchanged
        if (mx_out && threadIdx.x == 0) mx_out[(int64_t)b * PM_PARTS + blockIdx.x] = mx[(int64_t)b * PM_PARTS + blockIdx.x];
        return;
    }
    const double thresh = pair_max(mx, b) * 1e-8;
    const int64_t o = (int64_t)b * FF;
    const int w = p.w, M1 = 2 * w + 2;
    if (FUSED && tab) {
        // class tables (k_mask_tables): a pixel's six moments are six look-ups (L2-resident, 6 x (2w+3)^2 doubles per pair;
        // interior pixels -- nine in ten -- take the full-window entries from registers); nothing is staged, so the block
        // starts on its pixels at once.  The same values, operations and order as the per-pixel evaluation below.
        const int NC = 2 * w + 3;
        const double* T = tab + (int64_t)b * nmom * tab_stride;
        const MaskIv ix = mask_interval(F, w, p.flags & 1, p.flags & 2, 0, true);
        const MaskIv iy = mask_interval(F, w, p.flags & 4, p.flags & 8, 0, true);
        const int xf_lo = ix.lo + (ix.hlo ? 1 : 0), xf_hi = ix.hi - 2 * w - (ix.hhi ? 1 : 0);
        const int yf_lo = iy.lo + (iy.hlo ? 1 : 0), yf_hi = iy.hi - 2 * w - (iy.hhi ? 1 : 0);
        const int nq = bco == 1 ? 6 : 1, mid = (w + 1) * NC + (w + 1);
        double tot[6] = {0, 0, 0, 0, 0, 0};
        for (int q = 0; q < nq; ++q) tot[q] = T[q * tab_stride + mid];
        double m = -INFINITY;
        // a thread's pixels, PIX at a time with their P, xP, yP requested together (the plain loop waited for P, then for xP
        // and yP, pixel by pixel: twelve memory latencies in sequence at F = 256)
        constexpr int PIX = 4;
        const int stride = gridDim.x * blockDim.x;
        for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < FF; i0 += PIX * stride) {
            double Pv[PIX], xPv[PIX], yPv[PIX];
#pragma unroll
            for (int q = 0; q < PIX; ++q) {
                const int ii = min(i0 + q * stride, FF - 1);
                Pv[q] = A.P[o + ii];
                xPv[q] = bco == 1 ? A.xP[o + ii] : 0.0;
                yPv[q] = bco == 1 ? A.yP[o + ii] : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PIX; ++q) {
            const int i = i0 + q * stride;
            if (i >= FF) break;
            const double P = Pv[q];
            const int y = i / F, x = i - y * F;
            const int e = mask_class(y, yf_lo, yf_hi, w) * NC + mask_class(x, xf_lo, xf_hi, w);
            const bool full = e == mid;
            const double a00 = full ? tot[0] : T[e];
            if (!(a00 * P > thresh)) {
                m = fmax(m, P);
                continue;
            }
            const double normed = P / a00;
            if (bco == 0) {
                A.P[o + i] = normed;
                m = fmax(m, normed);
                continue;
            }
            double a10 = tot[1], a01 = tot[2], a20 = tot[3], a02 = tot[4], a11 = tot[5];
            if (!full)
                a10 = T[tab_stride + e], a01 = T[2 * tab_stride + e], a20 = T[3 * tab_stride + e], a02 = T[4 * tab_stride + e],
                a11 = T[5 * tab_stride + e];
            const double xP = xPv[q], yP = yPv[q];
            const double denom = a20 * (a01 * a01) + (a10 * a10) * a02 - a00 * a02 * a20 + (a11 * a11) * a00 - 2 * a01 * a10 * a11;
            const double Aq = a11 * a11 - a02 * a20;
            const double Ax = a10 * a02 - a01 * a11;
            const double Ay = a01 * a20 - a10 * a11;
            const double corrected = (P * Aq + xP * Ax + yP * Ay) / denom;
            const double out = normed * exp(fmin(corrected / normed, 4.0) - 1.0);
            A.P[o + i] = out;
            m = fmax(m, out);
            }
        }
        if (mx_out) {
            m = block_max(m, red);
            if (threadIdx.x == 0) mx_out[(int64_t)b * PM_PARTS + blockIdx.x] = m;
        }
        return;
    }
    // FUSED: the pair's tables come into LDS once per block (dynamic shared memory: nq * sat_stride doubles) -- a border
    // pixel reads up to nine rectangles of each of the six, and from global memory those reads were its critical path
    extern __shared__ double sat_sh[];
    const double* s0 = nullptr;
    const MaskIv ix = mask_interval(F, w, p.flags & 1, p.flags & 2, 0, true);
    const MaskIv iy = mask_interval(F, w, p.flags & 4, p.flags & 8, 0, true);
    double tot[6] = {0, 0, 0, 0, 0, 0};
    if (FUSED) {
        const int nq = bco == 1 ? 6 : 1;
        const double* sg = sat + (int64_t)b * nmom * sat_stride;
        s0 = sg;
        if (stage_lds) {  // (uniform; wide windows' tables stay in global memory)
            for (int64_t e = threadIdx.x; e < (int64_t)nq * sat_stride; e += blockDim.x) sat_sh[e] = sg[e];
            __syncthreads();
            s0 = sat_sh;
        }
        for (int q = 0; q < nq; ++q) tot[q] = sat_rect(s0 + q * sat_stride, M1, 0, 2 * w, 0, 2 * w);
    }
    // where mask_geom's `full` holds: a rectangle of pixels (see k_rows_inv); they skip the geometry
    const int xf_lo = ix.lo + (ix.hlo ? 1 : 0), xf_hi = ix.hi - 2 * w - (ix.hhi ? 1 : 0);
    const int yf_lo = iy.lo + (iy.hlo ? 1 : 0), yf_hi = iy.hi - 2 * w - (iy.hhi ? 1 : 0);
    double m = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < FF; i += gridDim.x * blockDim.x) {
        const double P = A.P[o + i];
        MaskGeom g;
        double a00;
        if (FUSED) {
            const int y = i / F, x = i - y * F;
            if (x >= xf_lo && x <= xf_hi && y >= yf_lo && y <= yf_hi) {
                g.full = true, g.any = true;
                a00 = tot[0];
            } else {
                g = mask_geom(F, w, ix, iy, x, y);
                a00 = g.full ? tot[0] : mask_moment(s0, M1, g);
            }
        } else {
            a00 = A.a00[o + i];
        }
        if (!(a00 * P > thresh)) {
            m = fmax(m, P);
            continue;
        }
        const double normed = P / a00;
        if (bco == 0) {
            A.P[o + i] = normed;
            m = fmax(m, normed);
            continue;
        }
        double a10, a01, a20, a02, a11;
        if (FUSED) {
            if (g.full) {
                a10 = tot[1], a01 = tot[2], a20 = tot[3], a02 = tot[4], a11 = tot[5];
            } else {
                a10 = mask_moment(s0 + 1 * sat_stride, M1, g), a01 = mask_moment(s0 + 2 * sat_stride, M1, g);
                a20 = mask_moment(s0 + 3 * sat_stride, M1, g), a02 = mask_moment(s0 + 4 * sat_stride, M1, g);
                a11 = mask_moment(s0 + 5 * sat_stride, M1, g);
            }
        } else {
            a10 = A.a10[o + i], a01 = A.a01[o + i], a20 = A.a20[o + i], a02 = A.a02[o + i], a11 = A.a11[o + i];
        }
        const double xP = A.xP[o + i], yP = A.yP[o + i];
        const double denom = a20 * (a01 * a01) + (a10 * a10) * a02 - a00 * a02 * a20 + (a11 * a11) * a00 - 2 * a01 * a10 * a11;
        const double Aq = a11 * a11 - a02 * a20;
        const double Ax = a10 * a02 - a01 * a11;
        const double Ay = a01 * a20 - a10 * a11;
        const double corrected = (P * Aq + xP * Ax + yP * Ay) / denom;
        const double out = normed * exp(fmin(corrected / normed, 4.0) - 1.0);
        A.P[o + i] = out;
        m = fmax(m, out);
    }
    if (mx_out) {
        m = block_max(m, red);
        if (threadIdx.x == 0) mx_out[(int64_t)b * PM_PARTS + blockIdx.x] = m;
    }
}

// box = hist / bins2D where bins2D > max*1e-8 (mcsamples.py:1969-1971), embedded at (w,w)
__global__ void k_fill_box(const D2Pair* __restrict__ pairs, const double* __restrict__ hist, const double* __restrict__ P,
                           const double* __restrict__ mx, int F, int S, double* __restrict__ frames) {
    const int w = pairs[blockIdx.y].w;
    const double thresh = pair_max(mx, blockIdx.y) * 1e-8;
    const int64_t o = (int64_t)blockIdx.y * F * F, oh = (int64_t)pairs[blockIdx.y].hidx * F * F;
    double* fr = frames + (int64_t)blockIdx.y * S * S;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S * S; e += gridDim.x * blockDim.x) {
        const int r = e / S - w, c = e % S - w;
        double v = 0;
        if (r >= 0 && r < F && c >= 0 && c < F) {
            const double h = hist[oh + (int64_t)r * F + c], p = P[o + (int64_t)r * F + c];
            v = (p > thresh) ? h / p : h;
        }
        fr[e] = v;
    }
}

// bins2D = bins2D * conv / a00
__global__ void k_mbc_update(double* __restrict__ P, const double* __restrict__ conv, const double* __restrict__ a00,
                             int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        P[i] = (P[i] * conv[i]) / a00[i];
}

__global__ void k_normalise(double* __restrict__ P, const double* __restrict__ mx, int FF, int* __restrict__ status) {
    const double m = pair_max(mx, blockIdx.y);
    if (blockIdx.x == 0 && threadIdx.x == 0) status[blockIdx.y] = (m == 0.0) ? GD_ERR_EMPTY : GD_OK;
    double* p = P + (int64_t)blockIdx.y * FF;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < FF; i += gridDim.x * blockDim.x)
        p[i] = (m == 0.0) ? 0.0 : p[i] / m;
}


// ---- explicit prior masks (mask_function, mcsamples.py:1907-1919) --------------------------------------------
// A user callback may carve any shape out of the (F+2w)^2 prior mask, so the box shortcut of k_mask_eval does not
// apply: the moments  a_pq = conv(mask, Win x^p y^q, 'valid')  are summed directly for the one pair concerned.
struct MaskOv {
    const double* d_mask_bc;      // (F+2w)^2 mask after _setEdgeMask2D (boundary-correction moments), or nullptr
    const double* d_mask_mbc;     // (F+2w)^2 mask after _setAllEdgeMask2D (bias-correction a00), or nullptr
    const unsigned char* d_zero;  // F^2 flags: 1 where the mask excludes the pixel (bool_mask, :1919), or nullptr
};

// kwin[(i1+w)*(2w+1) + (i2+w)] = Win[i1][i2] * i2^px * i1^py
__global__ void k_moment_window(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, int px, int py,
                                double* __restrict__ kwin) {
    const D2Pair p = pairs[0];
    const int M = 2 * p.w + 1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M * M; e += gridDim.x * blockDim.x) {
        const int i1 = e / M - p.w, i2 = e % M - p.w;
        double v = win_raw(p, i1, i2) / wsum[0];
        for (int q = 0; q < px; ++q) v = v * (double)i2;
        for (int q = 0; q < py; ++q) v = v * (double)i1;
        kwin[e] = v;
    }
}

// dst[y][x] = sum_{i1,i2} kwin[i1][i2] * mask[y + w - i1][x + w - i2]   (one pair)
__global__ void k_mask_moment_direct(const double* __restrict__ kwin, const double* __restrict__ mask, int F, int w,
                                     double* __restrict__ dst) {
    const int M = 2 * w + 1, Mp = F + 2 * w;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * F; e += gridDim.x * blockDim.x) {
        const int y = e / F, x = e % F;
        double acc = 0;
        for (int a = 0; a < M; ++a) {
            const double* mrow = mask + (int64_t)(y + 2 * w - a) * Mp + (x + 2 * w);
            const double* krow = kwin + a * M;
            for (int b = 0; b < M; ++b) acc = fma(krow[b], mrow[-b], acc);
        }
        dst[e] = acc;
    }
}

// bins2D = bins2D * conv / a00 outside the excluded region only (mcsamples.py:1973-1974)
__global__ void k_mbc_update_masked(double* __restrict__ P, const double* __restrict__ conv, const double* __restrict__ a00,
                                    const unsigned char* __restrict__ zero, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = P[i] * conv[i];
        P[i] = zero[i] ? v : v / a00[i];
    }
}
__global__ void k_zero_masked(double* __restrict__ P, const unsigned char* __restrict__ zero, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (zero[i]) P[i] = 0.0;
}

// ---- mean likelihoods (mcsamples.py:1886-1903, 2004-2006) ----------------------------------------------------
// t = likehist / L where L > 0 (else likehist)
__global__ void k_likes_div(const double* __restrict__ likehist, const double* __restrict__ L, int64_t n,
                            double* __restrict__ t) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        t[i] = (L[i] > 0) ? likehist[i] / L[i] : likehist[i];
}
// L = (L > 0) ? L2 * L : L2
__global__ void k_likes_mul(const double* __restrict__ L2, int64_t n, double* __restrict__ L) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        L[i] = (L[i] > 0) ? L2[i] * L[i] : L2[i];
}
// L = L / P0 where P0 > 1e-4 max(P0), else 0.  grid (blocks, B)
__global__ void k_likes_ratio(const double* __restrict__ P0, const double* __restrict__ mx, int FF, double* __restrict__ L) {
    const double thresh = 1e-4 * pair_max(mx, blockIdx.y);
    const int64_t o = (int64_t)blockIdx.y * FF;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < FF; i += gridDim.x * blockDim.x)
        L[o + i] = (P0[o + i] > thresh) ? L[o + i] / P0[o + i] : 0.0;
}

// Direct (summation) 2D convolution with the Gaussian window, used by the mean-likelihood path: a sum of non-negative
// terms has no cancellation noise, so exact zeros stay zero and tiny likelihood weights keep their relative accuracy
// (the reference's FFT route makes its `bin2Dlikes > 0` mask depend on the sign of rounding noise).
// src, dst: B x n0 x n1.  wrap != 0: circular over both axes (convolve.py:262-294), else zero outside ('same').
// Block = 64 columns x 4 row groups; each thread accumulates RY consecutive rows so one operand load feeds RY taps.
// The window is synthesised into LDS in column chunks.  grid (tiles, B)
#define CONV_RY 4
#define CONV_WIN_DOUBLES 4096
__global__ void __launch_bounds__(256) k_conv2d_direct(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum,
                                                       const double* __restrict__ src, int n0, int n1, int wrap,
                                                       double* __restrict__ dst) {
    __shared__ double win[CONV_WIN_DOUBLES];
    const D2Pair p = pairs[blockIdx.y];
    const double ws = wsum[blockIdx.y];
    const int w = p.w, M = 2 * w + 1;
    const int C = max(1, min(M, CONV_WIN_DOUBLES / M));  // window columns per chunk
    const int tiles_x = (n1 + 63) / 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x = (blockIdx.x % tiles_x) * 64 + tx;
    const int y0 = (blockIdx.x / tiles_x) * (4 * CONV_RY) + ty * CONV_RY;
    const double* s = src + (int64_t)blockIdx.y * n0 * n1;
    double acc[CONV_RY];
#pragma unroll
    for (int k = 0; k < CONV_RY; ++k) acc[k] = 0.0;
    const bool live = x < n1 && y0 < n0;
    for (int c0 = 0; c0 < M; c0 += C) {
        const int nc = min(C, M - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < M * nc; e += 256) {
            const int r = e / nc, cc = e % nc;
            win[r * C + cc] = win_raw(p, r - w, c0 + cc - w) / ws;
        }
        __syncthreads();
        if (!live) continue;
        for (int cc = 0; cc < nc; ++cc) {
            int xs = x - (c0 + cc - w);
            if (wrap) {
                xs %= n1;
                if (xs < 0) xs += n1;
            } else if (xs < 0 || xs >= n1) {
                continue;
            }
            for (int r = y0 - w; r <= y0 + CONV_RY - 1 + w; ++r) {
                int rr = r;
                if (wrap) {
                    rr %= n0;
                    if (rr < 0) rr += n0;
                } else if (rr < 0 || rr >= n0) {
                    continue;
                }
                const double v = s[(int64_t)rr * n1 + xs];
#pragma unroll
                for (int k = 0; k < CONV_RY; ++k) {
                    const int i1 = y0 + k - r;
                    if (i1 >= -w && i1 <= w) acc[k] = fma(win[(i1 + w) * C + cc], v, acc[k]);
                }
            }
        }
    }
    if (!live) return;
    double* d = dst + (int64_t)blockIdx.y * n0 * n1;
#pragma unroll
    for (int k = 0; k < CONV_RY; ++k)
        if (y0 + k < n0) d[(int64_t)(y0 + k) * n1 + x] = acc[k];
}

// ---- periodic axes (convolve.py:215-323): circular convolution on the folded (Ny x Nx) grid ------------------
// fold an F x F array onto the circular grid: drop the last column/row of a periodic axis and add it to the first
__global__ void k_fill_circ(const double* __restrict__ src, int F, int Ny, int Nx, double* __restrict__ frames) {
    const double* s = src + (int64_t)blockIdx.y * F * F;
    double* fr = frames + (int64_t)blockIdx.y * Ny * Nx;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Ny * Nx; e += gridDim.x * blockDim.x) {
        const int r = e / Nx, c = e % Nx;
        double v = s[(int64_t)r * F + c];
        const bool wrap_c = (Nx < F) && c == 0, wrap_r = (Ny < F) && r == 0;
        if (wrap_c) v += s[(int64_t)r * F + (F - 1)];
        if (wrap_r) v += s[(int64_t)(F - 1) * F + c];
        if (wrap_c && wrap_r) v += s[(int64_t)(F - 1) * F + (F - 1)];
        fr[e] = v;
    }
}

// window * x^px * y^py centred with wrap on the Ny x Nx circular grid (np.roll of the zero-padded kernel)
__global__ void k_fill_window_rect(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, int Ny, int Nx,
                                   int px, int py, double* __restrict__ frames) {
    const D2Pair p = pairs[blockIdx.y];
    double* fr = frames + (int64_t)blockIdx.y * Ny * Nx;
    const double ws = wsum[blockIdx.y];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Ny * Nx; e += gridDim.x * blockDim.x) {
        const int r = e / Nx, c = e % Nx;
        // hpad[:k,:k] = win then roll by -(k//2): window index i (0..2w) lands at (i - w) mod N
        double v = 0;
        const int i1a = r, i1b = r - Ny, i2a = c, i2b = c - Nx;  // candidates for the centred offsets
        for (int ua = 0; ua < 2; ++ua) {
            const int i1 = ua ? i1b : i1a;
            if (i1 < -p.w || i1 > p.w) continue;
            for (int ub = 0; ub < 2; ++ub) {
                const int i2 = ub ? i2b : i2a;
                if (i2 < -p.w || i2 > p.w) continue;
                double t = win_raw(p, i1, i2) / ws;
                for (int q = 0; q < px; ++q) t = t * (double)i2;
                for (int q = 0; q < py; ++q) t = t * (double)i1;
                v += t;
            }
        }
        fr[e] = v;
    }
}

// circular result (Ny x Nx) -> F x F with the first row/column repeated at the end of a periodic axis
__global__ void k_expand_circ(const double* __restrict__ frames, int F, int Ny, int Nx, double* __restrict__ dst) {
    const double* fr = frames + (int64_t)blockIdx.y * Ny * Nx;
    double* d = dst + (int64_t)blockIdx.y * F * F;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * F; e += gridDim.x * blockDim.x) {
        const int r = e / F, c = e % F;
        d[e] = fr[(int64_t)(r % Ny) * Nx + (c % Nx)];
    }
}

// box = hist / bins2D where bins2D > max*1e-8 else hist, as a plain F x F array
__global__ void k_box(const double* __restrict__ hist, const double* __restrict__ P, const double* __restrict__ mx, int FF,
                      double* __restrict__ box) {
    const double thresh = pair_max(mx, blockIdx.y) * 1e-8;
    const int64_t o = (int64_t)blockIdx.y * FF;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < FF; i += gridDim.x * blockDim.x) {
        const double h = hist[o + i], p = P[o + i];
        box[o + i] = (p > thresh) ? h / p : h;
    }
}

// ---- convolution in LDS: row transforms, column transform x window spectrum x inverse column transform, row transforms ----
// The rocFFT route moves every S x S frame through HBM about eight times per convolution (embed, two transform passes,
// spectral product, two inverse passes, crop: ~10.6 MB per pair at S = 288).  Here a convolution is three kernels and
// ~4 MB per pair:
//   k_rows_fwd   reads the F x F source (the histogram, or the bias-correction box formed on the fly), transforms each of
//                its F non-zero rows in LDS and writes the half spectrum transposed, Xt[b][kx][y]  (y < F only: the other
//                rows of the frame are zero);
//   k_col_conv   per column kx: transform along y, multiply by the window's spectrum -- built in the block from the
//                (2w+1)^2 window values: a direct sum over x for each window row, then the same column transform -- inverse
//                transform, keep the F rows of the crop: Yt[b][kx][y];
//   k_rows_inv   per output row: Hermitian-extend, inverse transform, crop, and the epilogue of k_crop_fused (bias update,
//                block maxima).
// A transform is a sequence of Stockham auto-sort passes of radix 3 / 5 / 4 / 2 done IN PLACE in one LDS buffer by
// FT = 32 lanes (two transforms per wavefront): every lane reads the inputs of all its butterflies of a pass into
// registers, the group synchronises, and the outputs go back to their auto-sorted places.  One buffer per transform is
// what lets ~30 transforms be resident per CU -- their number, not the arithmetic, bounds these kernels.  Any frame size of
// the ladder up to 512 is handled (larger grids keep the rocFFT route).
#include "ldsfft.hpp"
#include "fft288.hpp"

// Layout of the half spectra between the row and the column kernels (round 5): TILED by 16 rows,
//     T[b][y / 16][kx][y % 16]   (NT = ceil(F / 16) tiles of Sh x 16 complex values per pair),
// so that the 16 rows a row-kernel block transforms are ONE contiguous run of Sh * 256 bytes (37 KB at S = 288): the block
// transposes inside LDS (un-mixed spectra stay in the rows' own buffers, row stride H + 1 complex values so that the 16
// rows of one kx fall on different banks) and stores / loads its tile with consecutive lanes on consecutive addresses.
// Until round 4 the layout was [b][kx][y] and a 64-lane store of the row kernels touched 32 different cache lines with 32
// bytes each (lanes = 32 different kx, two rows).  The column kernels read / write a column as 256-byte segments of
// adjacent kx (a block carries 8-12 adjacent columns: 2-3 KB runs per tile).
#define XT_ROWS 16
__device__ __forceinline__ int64_t xt_at(int b, int NT, int Sh, int y, int kx) {
    return (((int64_t)b * NT + (y >> 4)) * Sh + kx) * XT_ROWS + (y & (XT_ROWS - 1));
}

// grid (ceil(F / 16), B), 16 * 32 threads = the 16 rows of one tile of pair b per block.  A row is real: its even and odd
// samples are packed into one complex sequence of half the frame length H = S / 2 (plH), transformed, and un-mixed into the
// S / 2 + 1 spectrum values (twg: the S twiddles, stride 2 for the transform) IN PLACE (X[0] and X[H] are real and share
// slot 0); after a block barrier the tile goes out in one contiguous run.
// MODE 0: rows of src (B x F x F); MODE 1: rows of the bias-correction box src / P where P > max * 1e-8 (k_fill_box).
// FTL lanes per row: 32 for the Stockham passes; 16 for the 144-point register transform (round 5) -- its 16-point step
// keeps 9 lanes of a row's group busy and its 9-point step 16, so with 32-lane groups a wave carried two rows with 18 and
// 32 of its 64 lanes at work; with 16-lane groups it carries four (36 and 64 lanes): the transform's instructions per row
// halve, everything else (pack, un-mix, tile store) is lane-parallel either way.  The arithmetic of a row is unchanged.
// FTL = 64 (round 6): the frames above 512 points (the up-scaled grid classes: half-length transforms of up to 576 points on
// a whole wavefront); TWG: the twiddles are read where they lie (global memory, cache-resident) -- the 16 row buffers of a
// 1152-point frame leave no room for the table in LDS.
template <int MODE, int FTL, bool TWG = false>
__global__ void __launch_bounds__(16 * FTL) k_rows_fwd(const D2Pair* __restrict__ pairs, const double* __restrict__ src,
                                                  const double* __restrict__ P, const double* __restrict__ mx, int F, FftDev plH,
                                                  const double2* __restrict__ twg, double2* __restrict__ Xt) {
    extern __shared__ double2 sh2[];
    __shared__ double thresh_sh;
    const int H = plH.S, S = 2 * H, Sh = H + 1, RP = H + 1, b = blockIdx.y;
    const double2* tw = TWG ? twg : sh2;
    double2* const rows0 = sh2 + (TWG ? 0 : S);
    const int g = threadIdx.x / FTL, t = threadIdx.x % FTL;
    double2* buf = rows0 + (size_t)g * RP;
    if (!TWG) {  // (both of a thread's twiddle loads requested before the first is stored)
        const int i0 = threadIdx.x, i1 = threadIdx.x + blockDim.x;
        const double2 a0 = twg[min(i0, S - 1)], a1 = twg[min(i1, S - 1)];
        if (i0 < S) sh2[i0] = a0;
        if (i1 < S) sh2[i1] = a1;
        for (int i = threadIdx.x + 2 * blockDim.x; i < S; i += blockDim.x) sh2[i] = twg[i];
    }
    if (MODE == 1 && threadIdx.x == 0) thresh_sh = pair_max(mx, b) * 1e-8;
    __syncthreads();
    const int w = pairs[b].w;
    const int y = blockIdx.x * XT_ROWS + g;
    const bool active = y < F;
    if (active) {
        const int64_t o = (int64_t)b * F * F + (int64_t)y * F, oh = (int64_t)pairs[b].hidx * F * F + (int64_t)y * F;
        // The frame's row at position x is the source row embedded at offset w.  ALL of a lane's loads are requested before
        // the first value is used (unconditional loads from clamped addresses, the padding selected afterwards): the loop
        // this replaces issued two predicated 8-byte loads per iteration and waited for them before the next -- nine
        // memory latencies in sequence per block, which is what the row kernels' time was made of (ISA reading, round 5).
        constexpr int MAXIT = (FTL == 32) ? 8 : 9;  // ceil(H / FTL): H = 144 on 16 lanes, H <= 256 on 32, H <= 576 on 64
        double hv[2 * MAXIT], pv[2 * MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * (t + it * FTL) + e - w;
                const int cc = min(max(c, 0), F - 1);
                hv[2 * it + e] = src[oh + cc];
                if (MODE == 1) pv[2 * it + e] = P[o + cc];
            }
        __builtin_amdgcn_sched_barrier(0);  // (keep the loads together: the scheduler would sink each to its use)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int n = t + it * FTL;
            double ve[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * n + e - w;
                double v = hv[2 * it + e];
                if (MODE == 1) {
                    const double pp = pv[2 * it + e];
                    if (pp > thresh_sh) v = v / pp;
                }
                ve[e] = (c >= 0 && c < F) ? v : 0.0;
            }
            if (n < H) buf[n] = make_double2(ve[0], ve[1]);
        }
    }
    group_sync();
    if (FTL == 16)  // 16 x 9 in registers, fft288.hpp (the host launches this form for H = 144 only)
        f288::fft144_group<false>(reinterpret_cast<f288::C2*>(buf), reinterpret_cast<const f288::C2*>(tw), t);
    else if (H == 144)  // (uniform)
        f288::fft144_group<false>(reinterpret_cast<f288::C2*>(buf), reinterpret_cast<const f288::C2*>(tw), t);
    else
        fft_full<0, false, (FTL == 64 ? 64 : FT)>(buf, tw, 2, plH, t);
    if (active) {  // un-mix in place: the lane that consumes (buf[k], buf[H - k]) writes (X[k], X[H - k]) there
        for (int k = t; 2 * k <= H; k += FTL) {
            if (k == 0) {
                const double2 z = buf[0];
                buf[0] = make_double2(z.x + z.y, z.x - z.y);  // (X[0], X[H]): both real
                continue;
            }
            const double2 zk = buf[k], zm = buf[H - k];
            // X[k] = ((zk + conj zm) - i w^k (zk - conj zm)) / 2,  w = e^{-2 pi i / S}
            const double ax = zk.x + zm.x, ay = zk.y - zm.y, bx = zk.x - zm.x, by = zk.y + zm.y;
            const double2 e = tw[k];
            const double u = fma(e.x, bx, -(e.y * by)), v = fma(e.x, by, e.y * bx);
            buf[k] = make_double2(0.5 * (ax + v), 0.5 * (ay - u));
            if (2 * k < H) {  // X[H - k] from the same two values (roles exchanged, w^(H - k))
                const double2 f = tw[H - k];
                const double u2 = -fma(f.x, bx, f.y * by), v2 = fma(f.x, by, -(f.y * bx));
                buf[H - k] = make_double2(0.5 * (ax + v2), 0.5 * (-ay - u2));
            }
        }
    }
    __syncthreads();
    // the tile [kx][16 rows] in one contiguous run (rows of the tile past F carry whatever their idle buffers held: the
    // column kernels never read them)
    const int NT = (F + XT_ROWS - 1) / XT_ROWS;
    double2* tile = Xt + ((int64_t)b * NT + blockIdx.x) * Sh * XT_ROWS;
    const double2* rows = rows0;
    for (int i = threadIdx.x; i < Sh * XT_ROWS; i += blockDim.x) {
        const int kx = i >> 4, row = i & (XT_ROWS - 1);
        const double2* rb = rows + (size_t)row * RP;
        double2 v;
        if (kx == 0)
            v = make_double2(rb[0].x, 0.0);
        else if (kx == H)
            v = make_double2(rb[0].y, 0.0);
        else
            v = rb[kx];
        tile[i] = v;
    }
}

// grid (ceil(Sh / 8), B), 256 threads = 8 columns per block.  The spectrum of the window moment Win * x^px * y^py, by
// columns: Wt[b][kx][ky].  Built from the (2w+1)^2 window values: a direct sum over x for each window row, then the column
// transform.  Once per pair and moment (the plain window serves the first convolution and the bias-correction round).
// SZ: size class of the transforms (ldsfft.hpp; 2 = the frames above 512 points on 64-lane groups, round 6); blockDim.x / lanes
// columns per block (8 where the LDS has room).  WGLOB: a window whose (2w+1)^2 values do not fit the LDS next to the
// transform buffers (w of 50+ bins) -- the sums over x come from k_win_rowdft (one thread per (window row, kx) instead of
// 2w+1 lanes with (2w+1)/2 terms each: a 193-bin window took 315 us per moment in the form above), the same terms in the
// same order, so the same spectrum bit for bit.
template <int SZ, bool WGLOB = false>
__global__ void __launch_bounds__(SZ == 2 ? 512 : 256) k_win_spec(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, FftDev pl,
                                                  const double2* __restrict__ twg, int px, int py, int maxw,
                                                  double* __restrict__ Wt, const double2* __restrict__ rowdft = nullptr) {
    extern __shared__ double2 sh2[];
    constexpr int FTN = SZ == 2 ? 64 : FT;
    const int S = pl.S, Sh = S / 2 + 1, b = blockIdx.y, ncol = blockDim.x / FTN;
    const D2Pair p = pairs[b];
    const int w = p.w, M = 2 * w + 1, Mmax = 2 * maxw + 1;
    double2* tw = sh2;
    double* wn = reinterpret_cast<double*>(sh2 + S);  // (2w+1)^2 window values, row stride M
    const int g = threadIdx.x / FTN, t = threadIdx.x % FTN;
    double2* bw = sh2 + S + (WGLOB ? 0 : (Mmax * Mmax + 1) / 2) + (size_t)g * S;
    for (int i = threadIdx.x; i < S; i += blockDim.x) tw[i] = twg[i];
    if (!WGLOB) {
        const double ws = wsum[b];
        for (int e = threadIdx.x; e < M * M; e += blockDim.x) {
            const int i1 = e / M - w, i2 = e % M - w;
            double v = win_raw(p, i1, i2) / ws;
            for (int q = 0; q < px; ++q) v = v * (double)i2;
            for (int q = 0; q < py; ++q) v = v * (double)i1;
            wn[e] = v;
        }
    }
    __syncthreads();
    const int kx = blockIdx.x * ncol + g;
    const bool active = kx < Sh;
    // rows wrapped into the frame; two lanes per row (the halves of its x range, added lower + upper on both lanes), so
    // that the group's lanes share the (2w+1)^2 terms instead of 2w+1 of them doing a row each
    for (int idx = t; idx < S; idx += FTN) bw[idx] = make_double2(0.0, 0.0);
    group_sync();
    if (WGLOB) {
        if (active) {
            const double2* rs = rowdft + (int64_t)b * Mmax * Sh + kx;  // row r of the window: + r * Sh
            for (int r = t; r < M; r += FTN) {
                const int i1 = r - w;
                bw[i1 >= 0 ? i1 : i1 + S] = rs[(int64_t)r * Sh];
            }
        }
    } else {
        const int half = (M + 1) / 2;
        for (int task0 = 0; task0 < 2 * M; task0 += FTN) {
            const int task = task0 + t, row_i = task >> 1, part = task & 1;
            double2 acc = make_double2(0.0, 0.0);
            if (active && row_i < M) {
                const double* row = wn + row_i * M;
                const int c0 = part ? half : 0, c1 = part ? M : half;
                // x offset c0 - w: e^{-2 pi i kx (c0 - w) / S}  (kx (S - w + c0) < 2 S^2: 32-bit)
                int ph = (int)(((unsigned int)kx * (unsigned int)(S - w + c0)) % (unsigned int)S);
                // four terms per trip with their eight LDS loads requested together (the plain loop waited for a load
                // pair per term: one LDS latency per term and wave); the sums keep their order term by term
                int c = c0;
                for (; c + 4 <= c1; c += 4) {
                    int p1 = ph + kx;
                    if (p1 >= S) p1 -= S;
                    int p2 = p1 + kx;
                    if (p2 >= S) p2 -= S;
                    int p3 = p2 + kx;
                    if (p3 >= S) p3 -= S;
                    const double r0 = row[c], r1 = row[c + 1], r2 = row[c + 2], r3 = row[c + 3];
                    const double2 e0 = tw[ph], e1 = tw[p1], e2 = tw[p2], e3 = tw[p3];
                    __builtin_amdgcn_sched_barrier(0);  // (or each load is sunk to its use again)
                    acc.x = fma(r0, e0.x, acc.x), acc.y = fma(r0, e0.y, acc.y);
                    acc.x = fma(r1, e1.x, acc.x), acc.y = fma(r1, e1.y, acc.y);
                    acc.x = fma(r2, e2.x, acc.x), acc.y = fma(r2, e2.y, acc.y);
                    acc.x = fma(r3, e3.x, acc.x), acc.y = fma(r3, e3.y, acc.y);
                    ph = p3 + kx;
                    if (ph >= S) ph -= S;
                }
                for (; c < c1; ++c) {
                    const double2 e = tw[ph];
                    acc.x = fma(row[c], e.x, acc.x), acc.y = fma(row[c], e.y, acc.y);
                    ph += kx;
                    if (ph >= S) ph -= S;
                }
            }
            const double ox = __shfl_xor(acc.x, 1), oy = __shfl_xor(acc.y, 1);
            if (active && row_i < M && part == 0) {
                const int i1 = row_i - w;
                bw[i1 >= 0 ? i1 : i1 + S] = make_double2(acc.x + ox, acc.y + oy);
            }
        }
    }
    group_sync();
    // (uniform) 16 x M in registers, fft288.hpp
    if (SZ == 0 && S == 288)
        f288::fft16_group<18, false>(reinterpret_cast<f288::C2*>(bw), reinterpret_cast<const f288::C2*>(tw), t);
    else if (SZ == 0 && S == 320)
        f288::fft16_group<20, false>(reinterpret_cast<f288::C2*>(bw), reinterpret_cast<const f288::C2*>(tw), t);
    else if (SZ == 1 && S == 384)
        f288::fft16_group<24, false>(reinterpret_cast<f288::C2*>(bw), reinterpret_cast<const f288::C2*>(tw), t);
    else
        fft_full<SZ, false, FTN>(bw, tw, 1, pl, t);
    if (active) {
        // the window is even, so the spectrum of an even moment (px + py even) is real and that of an odd one imaginary:
        // one double per entry, the other part is rounding noise of the sums
        double* col = Wt + ((int64_t)b * Sh + kx) * S;
        const bool odd = (px + py) & 1;
        for (int idx = t; idx < S; idx += FTN) col[idx] = odd ? bw[idx].y : bw[idx].x;
    }
}

// The wide windows' sums over x (k_win_spec<SZ, true>), in two small launches.
// k_win_table, grid (blocks, B): the (2w+1)^2 values of the window moment Win * x^px * y^py of every pair, row stride 2w + 1,
// pair stride (2 maxw + 1)^2 -- k_win_spec's own expressions.
// k_win_rowdft, grid (2 maxw + 1, ceil(Sh / 256), B), 256 threads = 256 adjacent kx of one window row:
//     rowdft[b][r][kx] = sum_c wn[r][c] e^{-2 pi i kx (c - w) / S}
// as k_win_spec's two lanes of a row form it: the lower and the upper half of the row each summed in order, then added.
__global__ void __launch_bounds__(256) k_win_table(const D2Pair* __restrict__ pairs, const double* __restrict__ wsum, int px, int py,
                                                   int maxw, double* __restrict__ wtab) {
    const int b = blockIdx.y;
    const D2Pair p = pairs[b];
    const int w = p.w, M = 2 * w + 1, Mmax = 2 * maxw + 1;
    const double ws = wsum[b];
    double* wn = wtab + (int64_t)b * Mmax * Mmax;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < M * M; e += gridDim.x * 256) {
        const int i1 = e / M - w, i2 = e % M - w;
        double v = win_raw(p, i1, i2) / ws;
        for (int q = 0; q < px; ++q) v = v * (double)i2;
        for (int q = 0; q < py; ++q) v = v * (double)i1;
        wn[e] = v;
    }
}
__global__ void __launch_bounds__(256) k_win_rowdft(const D2Pair* __restrict__ pairs, const double* __restrict__ wtab, int S,
                                                    const double2* __restrict__ twg, int maxw, double2* __restrict__ rowdft) {
    extern __shared__ double2 sh2[];  // the S twiddles
    const int b = blockIdx.z, r = blockIdx.x, Sh = S / 2 + 1, Mmax = 2 * maxw + 1;
    const int w = pairs[b].w, M = 2 * w + 1;
    if (r >= M) return;  // (uniform: a pair with a narrower window than the batch's widest)
    for (int i = threadIdx.x; i < S; i += 256) sh2[i] = twg[i];
    __syncthreads();
    const int kx = blockIdx.y * 256 + threadIdx.x;
    if (kx >= Sh) return;
    const double* row = wtab + (int64_t)b * Mmax * Mmax + (int64_t)r * M;
    const int half = (M + 1) / 2;
    double2 lo = make_double2(0.0, 0.0), hi = make_double2(0.0, 0.0);
    int pl = (int)(((int64_t)kx * (S - w)) % S), ph = (int)(((int64_t)kx * (S - w + half)) % S);
    for (int c = 0; c < half; ++c) {  // (the upper half has `half` or `half - 1` terms)
        const double2 e = sh2[pl];
        lo.x = fma(row[c], e.x, lo.x), lo.y = fma(row[c], e.y, lo.y);
        pl += kx;
        if (pl >= S) pl -= S;
        if (half + c < M) {
            const double2 f = sh2[ph];
            hi.x = fma(row[half + c], f.x, hi.x), hi.y = fma(row[half + c], f.y, hi.y);
            ph += kx;
            if (ph >= S) ph -= S;
        }
    }
    rowdft[((int64_t)b * Mmax + r) * Sh + kx] = make_double2(lo.x + hi.x, lo.y + hi.y);
}

// grid (ceil(Sh / 8), B), 256 threads = 8 columns per block, one LDS buffer per column: transform the source's column,
// multiply by the window's spectrum in the last pass (scaled), inverse transform, keep the F rows of the crop.
template <int SZ>
__global__ void __launch_bounds__(SZ == 2 ? 512 : 256) k_col_conv(const D2Pair* __restrict__ pairs, int F, FftDev pl, const double2* __restrict__ twg,
                                                  const double* __restrict__ Wt, int w_odd, const double2* __restrict__ Xt,
                                                  double2* __restrict__ Yt) {
    extern __shared__ double2 sh2[];
    constexpr int FTN = SZ == 2 ? 64 : FT;  // (SZ = 2: the frames above 512 points, blockDim.x / 64 columns per block)
    const int S = pl.S, Sh = S / 2 + 1, b = blockIdx.y, ncol = blockDim.x / FTN;
    const int w = pairs[b].w;
    double2* tw = sh2;
    const int g = threadIdx.x / FTN, t = threadIdx.x % FTN;
    double2* bh = sh2 + S + (size_t)g * S;
    for (int i = threadIdx.x; i < S; i += blockDim.x) tw[i] = twg[i];
    __syncthreads();
    const int kx = blockIdx.x * ncol + g;
    const bool active = kx < Sh;
    const int NT = (F + XT_ROWS - 1) / XT_ROWS;
    if (active) {
        for (int idx = t; idx < S; idx += FTN) {
            const int r = idx - w;
            bh[idx] = (r >= 0 && r < F) ? Xt[xt_at(b, NT, Sh, r, kx)] : make_double2(0.0, 0.0);
        }
    }
    group_sync();
    const int Ns = fft_head<SZ, false, FTN>(bh, tw, 1, pl, t);
    const double scale = 1.0 / ((double)S * (double)S);
    const double* wcol = Wt + ((int64_t)b * Sh + (active ? kx : 0)) * S;  // real part (even moment) or imaginary part (odd)
    fft_pass_any<SZ, false, FTN>(pl, pl.nst - 1, bh, tw, 1, Ns, t, [&](int pos, double2 v) {
        const double ws = wcol[pos] * scale;
        bh[pos] = w_odd ? make_double2(-v.y * ws, v.x * ws) : make_double2(v.x * ws, v.y * ws);
    });
    fft_full<SZ, true, FTN>(bh, tw, 1, pl, t);
    if (active) {
        for (int r = t; r < F; r += FTN) Yt[xt_at(b, NT, Sh, r, kx)] = bh[r + w];
    }
}

// The same column convolution for S = 16 M, M = 18 / 20 / 24 (288: F = 256 with windows up to 16 bins, the triangle's main
// class; 320 and 384: wider windows) on the 16 x M register transforms of fft288.hpp.  grid (ceil(Sh / cols per block), B),
// 192 threads: a wave carries 64 / M columns -- lane M c + n2 does the 16-point transforms of column c (forward: straight
// from the source column in global memory, zero padding included), lane 16 c + k1 the M-point ones (the inverse's straight
// into the cropped output).  Six LDS accesses per value where the radix passes make forty (measured on 136 pairs of
// 288-point frames, `scripts/r04_conv_bench.py`: 43.7 against 89.7 us per launch; the C3 step 27.3 against 28.0 ms).
template <int M>
__global__ void __launch_bounds__(192) k_col_conv16(const D2Pair* __restrict__ pairs, int F, const double2* __restrict__ twg,
                                                    const double* __restrict__ Wt, int w_odd, const double2* __restrict__ Xt,
                                                    double2* __restrict__ Yt) {
    using namespace f288;
    extern __shared__ double2 sh2[];
    constexpr int S = 16 * M, Sh = S / 2 + 1, CW = 64 / M, COLS = 3 * CW;
    const int b = blockIdx.y, w = pairs[b].w;
    C2* tw = reinterpret_cast<C2*>(sh2);
    for (int i = threadIdx.x; i < S; i += 192) sh2[i] = twg[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c1 = lane / M, n2 = lane - M * c1, c2 = lane >> 4, k1 = lane & 15;
    const int kx1 = blockIdx.x * COLS + wave * CW + c1, kx2 = blockIdx.x * COLS + wave * CW + c2;
    const bool act1 = c1 < CW && kx1 < Sh, act2 = c2 < CW && kx2 < Sh;
    C2* buf1 = reinterpret_cast<C2*>(sh2) + S + (size_t)(wave * CW + (c1 < CW ? c1 : 0)) * S;
    C2* buf2 = reinterpret_cast<C2*>(sh2) + S + (size_t)(wave * CW + (c2 < CW ? c2 : 0)) * S;
    C2 v[M];
    // ---- forward, 16-point transforms over n1 of x[M n1 + n2]; the frame's row p holds source row p - w
    const int NT = (F + XT_ROWS - 1) / XT_ROWS;
    if (act1) {
        const double2* col = Xt + xt_at(b, NT, Sh, 0, kx1);  // row r of the column: + (r / 16) * Sh * 16 + r % 16
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int r = M * n1 + n2 - w;
            const double2 x = (r >= 0 && r < F) ? col[(int64_t)(r >> 4) * (Sh * XT_ROWS) + (r & (XT_ROWS - 1))] : make_double2(0.0, 0.0);
            v[n1] = C2{x.x, x.y};
        }
        dft16<false>(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            C2 a = v[dft16_at(q)];
            if (q > 0) {
                const C2 e = tw[n2 * q];  // e^{-2 pi i n2 q / S} = (cos, -sin)
                a = rot<false>(a, e.x, -e.y);
            }
            buf1[M * q + n2] = a;
        }
    }
    group_sync();
    // ---- forward, M-point transforms over n2; times the window's spectrum (real or imaginary, scaled)
    if (act2) {
        const double* wcol = Wt + ((int64_t)b * Sh + kx2) * S;
        double ws[M];
#pragma unroll
        for (int q = 0; q < M; ++q) ws[q] = wcol[k1 + 16 * q];
#pragma unroll
        for (int q = 0; q < M; ++q) v[q] = buf2[M * k1 + q];
        Second<M>::template run<false>(v);
        const double scale = 1.0 / ((double)S * (double)S);
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const double m = ws[q] * scale;
            const C2 a = v[Second<M>::at(q)];
            v[Second<M>::at(q)] = w_odd ? C2{-a.y * m, a.x * m} : C2{a.x * m, a.y * m};
        }
    }
    group_sync();  // every lane of the wave has read its M values
    if (act2) {
#pragma unroll
        for (int q = 0; q < M; ++q) buf2[k1 + 16 * q] = v[Second<M>::at(q)];
    }
    group_sync();
    // ---- inverse, 16-point transforms
    if (act1) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = buf1[M * n1 + n2];
        dft16<true>(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            C2 a = v[dft16_at(q)];
            if (q > 0) {
                const C2 e = tw[n2 * q];
                a = rot<true>(a, e.x, -e.y);
            }
            buf1[M * q + n2] = a;
        }
    }
    group_sync();
    // ---- inverse, M-point transforms; the crop rows go straight out
    if (act2) {
#pragma unroll
        for (int q = 0; q < M; ++q) v[q] = buf2[M * k1 + q];
        Second<M>::template run<true>(v);
        double2* col = Yt + xt_at(b, NT, Sh, 0, kx2);
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const int r = k1 + 16 * q - w;
            const C2 a = v[Second<M>::at(q)];
            if (r >= 0 && r < F) col[(int64_t)(r >> 4) * (Sh * XT_ROWS) + (r & (XT_ROWS - 1))] = make_double2(a.x, a.y);
        }
    }
}

// grid (ceil(F / RPB), B), RPB * 32 threads.  The inverse of k_rows_fwd's packing: the S / 2 + 1 spectrum values of a real
// row are mixed into the half-length complex sequence whose inverse transform carries the row's even samples in its real
// and the odd ones in its imaginary part (unnormalised like a length-S inverse).  MODE 0: dst = crop; MODE 1: dst = dst *
// crop / a00 (the multiplicative bias-correction update); mx (may be nullptr): block maxima of what was written, parts the
// grid does not cover set to -inf.
// MODE 1 takes the divisor a00 (the all-edge mask's zeroth moment) from its F x F array, or -- sat1 given: the pair's
// summed-area table of that moment at sat1 + b * sat1_pair_stride -- evaluates it per pixel (k_mask_eval's operations).
template <int MODE, int FTL, bool TWG = false>
__global__ void __launch_bounds__(16 * FTL) k_rows_inv(const D2Pair* __restrict__ pairs, const double2* __restrict__ Yt, int F, FftDev plH,
                                                  const double2* __restrict__ twg, double* __restrict__ dst,
                                                  const double* __restrict__ a00, double* __restrict__ mx,
                                                  const double* __restrict__ sat1, int64_t sat1_pair_stride, int edge_applied,
                                                  int sat_in_lds, const double* __restrict__ tab1 = nullptr,
                                                  int64_t tab1_pair_stride = 0) {
    extern __shared__ double2 sh2[];
    __shared__ double red[16];
    const int H = plH.S, S = 2 * H, Sh = H + 1, RP = H + 1, b = blockIdx.y;
    const double2* tw = TWG ? twg : sh2;
    double2* const rows0 = sh2 + (TWG ? 0 : S);
    const int g = threadIdx.x / FTL, t = threadIdx.x % FTL;
    double2* buf = rows0 + (size_t)g * RP;
    {  // the block's tile of Yt, one contiguous run, into the rows' buffers (row stride H + 1: slot kx of row r); the twiddles.
        // A thread's loads of a batch are all requested before the first is stored (the plain loop waited for each one).
        const int NT = (F + XT_ROWS - 1) / XT_ROWS;
        const double2* tile = Yt + ((int64_t)b * NT + blockIdx.x) * Sh * XT_ROWS;
        double2* rows = rows0;
        const int total = Sh * XT_ROWS;
        constexpr int BATCH = 5;
        for (int i0 = threadIdx.x; i0 < total; i0 += BATCH * blockDim.x) {
            double2 v[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) v[q] = tile[min(i0 + q * (int)blockDim.x, total - 1)];
            __builtin_amdgcn_sched_barrier(0);  // (or the scheduler sinks every load to its store again)
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {  // (unconditional: past the end the last element is stored again, by design --
                // a predicated store gets its load sunk into its own basic block and waited for there)
                const int i = min(i0 + q * (int)blockDim.x, total - 1);
                rows[(size_t)(i & (XT_ROWS - 1)) * RP + (i >> 4)] = v[q];
            }
        }
        if (!TWG) {
            const int i0 = threadIdx.x, i1 = threadIdx.x + blockDim.x;
            const double2 a0 = twg[min(i0, S - 1)], a1 = twg[min(i1, S - 1)];
            if (i0 < S) sh2[i0] = a0;
            if (i1 < S) sh2[i1] = a1;
            for (int i = threadIdx.x + 2 * blockDim.x; i < S; i += blockDim.x) sh2[i] = twg[i];
        }
    }
    const int w = pairs[b].w;
    // MODE 1 with tables: the pair's summed-area table of the divisor comes into LDS once -- the border pixels of every row
    // read a dozen of its entries each, and from global memory each such read sat on the row's critical path
    // (measured with the interior shortcut below: 141 -> 107 us per 136-pair launch; requesting the row of the grid ahead of the
    // transform instead of at its use: 117, not kept)
    double* sat_l = reinterpret_cast<double*>(rows0 + (size_t)XT_ROWS * RP);
    // MODE 1 with class tables (k_mask_tables): a row needs the 2w + 3 entries of its y class only -- its group fetches
    // them (LDS: 16 x (2w + 3) doubles) and a pixel's divisor is one look-up
    if (MODE == 1 && tab1) {
        const int fl = pairs[b].flags, NC = 2 * w + 3, yy = blockIdx.x * XT_ROWS + g;
        const MaskIv iy = mask_interval(F, w, fl & 4, fl & 8, 1, edge_applied != 0);
        const int cy = mask_class(min(yy, F - 1), iy.lo + (iy.hlo ? 1 : 0), iy.hi - 2 * w - (iy.hhi ? 1 : 0), w);
        const double* trow = tab1 + (int64_t)b * tab1_pair_stride + (int64_t)cy * NC;
        for (int c = t; c < NC; c += FTL) sat_l[g * NC + c] = trow[c];
    } else if (MODE == 1 && sat1 && sat_in_lds) {
        const double* s1g = sat1 + (int64_t)b * sat1_pair_stride;
        const int n1 = (2 * w + 2) * (2 * w + 2);
        for (int i = threadIdx.x; i < n1; i += blockDim.x) sat_l[i] = s1g[i];
    }
    __syncthreads();
    const int y = blockIdx.x * XT_ROWS + g;
    const bool active = y < F;
    if (active) {  // in place: the lane that consumes (X[k], X[H - k]) writes (Z[k], Z[H - k]) there
        for (int k = t; 2 * k <= H; k += FTL) {
            if (k == 0) {
                const double x0 = buf[0].x, xh = buf[H].x;  // the imaginary parts of X[0], X[H] do not enter
                buf[0] = make_double2(x0 + xh, x0 - xh);
                continue;
            }
            const double2 xk = buf[k], xm = buf[H - k];
            // Z[k] = (xk + conj xm) + i conj(w^k) (xk - conj xm)
            const double ax = xk.x + xm.x, ay = xk.y - xm.y, bx = xk.x - xm.x, by = xk.y + xm.y;
            const double2 e = tw[k];  // w^k = (e.x, e.y); conj: (e.x, -e.y)
            const double u = fma(e.x, bx, e.y * by), v = fma(e.x, by, -(e.y * bx));  // conj(w^k) * (bx, by)
            buf[k] = make_double2(ax - v, ay + u);
            if (2 * k < H) {
                const double2 f = tw[H - k];
                const double u2 = fma(f.y, by, -(f.x * bx)), v2 = fma(f.x, by, f.y * bx);
                buf[H - k] = make_double2(ax - v2, -ay + u2);
            }
        }
    }
    group_sync();
    if (FTL == 16 || H == 144)
        f288::fft144_group<true>(reinterpret_cast<f288::C2*>(buf), reinterpret_cast<const f288::C2*>(tw), t);
    else
        fft_full<0, true, (FTL == 64 ? 64 : FT)>(buf, tw, 2, plH, t);
    double m = -INFINITY;
    if (active) {
        const int64_t o = (int64_t)b * F * F + (int64_t)y * F;
        const int fl = pairs[b].flags, M1 = 2 * w + 2;
        const double* s1 = (MODE == 1 && sat1) ? (sat_in_lds ? sat_l : sat1 + (int64_t)b * sat1_pair_stride) : nullptr;
        const MaskIv ix = mask_interval(F, w, fl & 1, fl & 2, 1, edge_applied != 0);
        const MaskIv iy = mask_interval(F, w, fl & 4, fl & 8, 1, edge_applied != 0);
        const double tot = (s1 && !tab1) ? sat_rect(s1, M1, 0, 2 * w, 0, 2 * w) : 0.0;
        // where mask_geom's `full` holds (the whole window inside the mask, no half row or column): a rectangle of pixels,
        // known per row -- the interior pixels skip the geometry altogether
        const int xf_lo = ix.lo + (ix.hlo ? 1 : 0), xf_hi = ix.hi - 2 * w - (ix.hhi ? 1 : 0);
        const bool row_full = y >= iy.lo + (iy.hlo ? 1 : 0) && y <= iy.hi - 2 * w - (iy.hhi ? 1 : 0);
        const double* drow = sat_l + g * (2 * w + 3);
        // (MODE 1: the row of the grid that is updated is requested in one go -- up to MAXX loads in flight per lane -- instead
        // of one load, waited for, per pixel)
        constexpr int MAXX = 16;
        double dv[MAXX];
        if (MODE == 1) {
#pragma unroll
            for (int it = 0; it < MAXX; ++it) dv[it] = dst[o + min(t + it * FTL, F - 1)];
            __builtin_amdgcn_sched_barrier(0);
        }
        auto pixel = [&](int x, double old) {
            const int pos = x + w;
            const double2 z = buf[pos >> 1];
            double v = (pos & 1) ? z.y : z.x;
            if (MODE == 1) {
                double div;
                if (tab1) {
                    div = drow[mask_class(x, xf_lo, xf_hi, w)];
                } else if (s1) {
                    if (row_full && x >= xf_lo && x <= xf_hi) {
                        div = tot;
                    } else {
                        const MaskGeom g = mask_geom(F, w, ix, iy, x, y);
                        div = g.full ? tot : mask_moment(s1, M1, g);
                    }
                } else {
                    div = a00[o + x];
                }
                v = (old * v) / div;
            }
            dst[o + x] = v;
            m = fmax(m, v);
        };
#pragma unroll
        for (int it = 0; it < MAXX; ++it) {
            const int x = t + it * FTL;
            if (x < F) pixel(x, MODE == 1 ? dv[it] : 0.0);
        }
        for (int x = t + MAXX * FTL; x < F; x += FTL) pixel(x, MODE == 1 ? dst[o + x] : 0.0);
    }
    if (mx) {
        m = block_max(m, red);
        if (threadIdx.x == 0) mx[(int64_t)b * PM_PARTS + blockIdx.x] = m;
        if (blockIdx.x == 0)
            for (int part = gridDim.x + threadIdx.x; part < PM_PARTS; part += blockDim.x) mx[(int64_t)b * PM_PARTS + part] = -INFINITY;
    }
}

static int next_fft_size(int n) {
    // smallest 2^a * {1,3,5,9,15} (a >= 4) >= n: a coarse ladder (288, 320, 384, 480, 512, 576, 640, 768, ...) keeps
    // the number of distinct plans small; any zero padding gives the same linear convolution
    int best = 1 << 30;
    const int odd[5] = {1, 3, 5, 9, 15};
    for (int a = 4; a < 28; ++a)
        for (int q = 0; q < 5; ++q) {
            const long long v = (1LL << a) * odd[q];
            if (v >= n && v < best) best = (int)v;
        }
    return best;
}

// Periodic variant (one or both axes periodic, the same for the whole batch).  Histogram-side convolutions are
// circular on the folded (Ny x Nx) grid -- including along a non-periodic axis, which the reference also wraps
// (convolve.py:226-251 takes the FFT at the array size without zero-padding); mask-side 'valid' convolutions use the
// ordinary zero-padded frames.  mcsamples.py:1874-1976.
// With an explicit prior mask (`ov`, one pair: a user's mask_function, mcsamples.py:1907-1919) the mask moments are summed
// directly over the (F + 2w)^2 array the caller edited -- 'valid' and not circular, exactly as the reference convolves
// the mask (:1924-1951, :1967) while the histogram side stays circular -- and the excluded region is exempted from the
// a00 division (:1973-1976) and zeroed at the end (:1978-1979).
static int density2d_periodic(gd_ctx* ctx, int B, int F, const double* d_hist, const std::vector<D2Pair>& hp, int maxw,
                              int per, int bco, int mbc, double* d_P, int32_t* status_out, bool wait,
                              const MaskOv* ov = nullptr) {
    GD_REQUIRE(!ov || B == 1, "an explicit prior mask belongs to one pair");
    const bool px = per & 16, py = per & 32, both = px && py;
    const int Nx = px ? F - 1 : F, Ny = py ? F - 1 : F;
    GD_REQUIRE(2 * maxw + 1 <= Nx && 2 * maxw + 1 <= Ny, "window wider than the periodic grid");
    bool any_prior = false;
    for (int b = 0; b < B; ++b) any_prior |= (hp[b].flags & 64) != 0;
    const bool do_bc = any_prior && bco >= 0 && !both;
    const bool do_mbc = mbc > 0 && !both;
    const int S = next_fft_size(F + 2 * maxw), Sh = S / 2 + 1, Nxh = Nx / 2 + 1;
    const int64_t FF = (int64_t)F * F, SS = (int64_t)S * S, SC = (int64_t)S * Sh, NN = (int64_t)Ny * Nx,
                  NC = (int64_t)Ny * Nxh;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const bool need_S = (do_bc || do_mbc) && !ov;
    const int64_t o_kwin = take(ov ? (int64_t)(2 * maxw + 1) * (2 * maxw + 1) * 8 : 0);
    const int64_t o_pairs = take((int64_t)B * sizeof(D2Pair)), o_wsum = take((int64_t)B * 8), o_mx = take((int64_t)B * 8 * PM_PARTS),
                  o_status = take((int64_t)B * 4), o_RC = take(B * NN * 8), o_ROc = take(B * NN * 8),
                  o_ZHc = take(B * NC * 16), o_ZWc = take(B * NC * 16), o_ZKc = take(B * NC * 16),
                  o_ZPc = take(B * NC * 16), o_RF = take(need_S ? B * SS * 8 : 0), o_RO = take(need_S ? B * SS * 8 : 0),
                  o_ZW = take(need_S ? B * SC * 16 : 0), o_ZM = take(need_S ? B * SC * 16 : 0),
                  o_ZK = take(need_S ? B * SC * 16 : 0), o_ZP = take(need_S ? B * SC * 16 : 0),
                  o_arr = take((do_bc ? (bco == 1 ? 8 : 1) : 0) * B * FF * 8), o_a00m = take(do_mbc ? B * FF * 8 : 0),
                  o_conv = take(B * FF * 8), o_box = take(do_mbc ? B * FF * 8 : 0);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    D2Pair* d_pairs = (D2Pair*)(base + o_pairs);
    double* d_wsum = (double*)(base + o_wsum);
    double* d_mx = (double*)(base + o_mx);
    int* d_status = (int*)(base + o_status);
    double* d_kwin = (double*)(base + o_kwin);
    // a mask moment of the explicit mask: dst = conv(mask, Win x^px y^py, 'valid')
    auto direct_moment = [&](const double* mask, int px_, int py_, double* dst) -> int {
        if (!mask) return gd_fail(ctx, GD_ERR_BADARG, "explicit mask missing for a requested correction");
        k_moment_window<<<64, 256, 0, ctx->stream>>>(d_pairs, d_wsum, px_, py_, d_kwin);
        if (hipGetLastError() != hipSuccess) return gd_fail(ctx, GD_ERR_HIP, "k_moment_window launch failed");
        k_mask_moment_direct<<<1024, 256, 0, ctx->stream>>>(d_kwin, mask, F, maxw, dst);
        if (hipGetLastError() != hipSuccess) return gd_fail(ctx, GD_ERR_HIP, "k_mask_moment_direct launch failed");
        return GD_OK;
    };
    double *RC = (double*)(base + o_RC), *ROc = (double*)(base + o_ROc), *RF = (double*)(base + o_RF),
           *RO = (double*)(base + o_RO), *arr = (double*)(base + o_arr), *d_a00m = (double*)(base + o_a00m),
           *d_conv = (double*)(base + o_conv), *d_box = (double*)(base + o_box);
    double2 *ZHc = (double2*)(base + o_ZHc), *ZWc = (double2*)(base + o_ZWc), *ZKc = (double2*)(base + o_ZKc),
            *ZPc = (double2*)(base + o_ZPc), *ZW = (double2*)(base + o_ZW), *ZM = (double2*)(base + o_ZM),
            *ZK = (double2*)(base + o_ZK), *ZP = (double2*)(base + o_ZP);
    if (wait) {
        GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(D2Pair)));
    } else {  // hp dies with this frame before the copy executes
        const int rc_stage = gd_stage_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(D2Pair));
        if (rc_stage) return rc_stage;
    }
    const dim3 gS(128, B), gF(64, B), gN(64, B);
    const double scaleS = 1.0 / ((double)S * (double)S), scaleN = 1.0 / ((double)Ny * (double)Nx);
    int rc;
    // circular convolution of a folded operand spectrum with a window spectrum -> F x F
    auto circ_conv = [&](double2* ZA, double2* ZB, double* dst) -> int {
        k_cmul<<<1024, 256, 0, ctx->stream>>>(ZA, ZB, B * NC, scaleN, ZPc);
        if (hipGetLastError() != hipSuccess) return gd_fail(ctx, GD_ERR_HIP, "k_cmul launch failed");
        int r = gd_fft_c2r_2d(ctx, Ny, Nx, B, ZPc, ROc);
        if (r) return r;
        k_expand_circ<<<gF, 256, 0, ctx->stream>>>(ROc, F, Ny, Nx, dst);
        return GD_OK;
    };
    auto mask_conv = [&](double2* ZA, double2* ZB, double* dst) -> int {
        k_cmul<<<1024, 256, 0, ctx->stream>>>(ZA, ZB, B * SC, scaleS, ZP);
        if (hipGetLastError() != hipSuccess) return gd_fail(ctx, GD_ERR_HIP, "k_cmul launch failed");
        int r = gd_fft_c2r_2d(ctx, S, S, B, ZP, RO);
        if (r) return r;
        k_crop<<<gF, 256, 0, ctx->stream>>>(d_pairs, RO, F, S, dst);
        return GD_OK;
    };
    k_win_sum<<<B, 256, 0, ctx->stream>>>(d_pairs, d_wsum);
    GD_KERNEL_CHECK();
    k_fill_window_rect<<<gN, 256, 0, ctx->stream>>>(d_pairs, d_wsum, Ny, Nx, 0, 0, RC);
    GD_KERNEL_CHECK();
    if ((rc = gd_fft_r2c_2d(ctx, Ny, Nx, B, RC, ZWc))) return rc;
    k_fill_circ<<<gN, 256, 0, ctx->stream>>>(d_hist, F, Ny, Nx, RC);
    GD_KERNEL_CHECK();
    if ((rc = gd_fft_r2c_2d(ctx, Ny, Nx, B, RC, ZHc))) return rc;
    if ((rc = circ_conv(ZHc, ZWc, d_P))) return rc;
    if (need_S) {
        k_fill_window<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_wsum, S, 0, 0, RF);
        GD_KERNEL_CHECK();
        if ((rc = gd_fft_r2c_2d(ctx, S, S, B, RF, ZW))) return rc;
    }
    if (do_bc) {
        BcArrays A;
        A.P = d_P;
        A.a00 = arr;
        A.a10 = A.a01 = A.a20 = A.a02 = A.a11 = A.xP = A.yP = nullptr;
        k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_P, (int)FF, d_mx);
        GD_KERNEL_CHECK();
        if (ov) {
            if ((rc = direct_moment(ov->d_mask_bc, 0, 0, A.a00))) return rc;
        } else {
            k_fill_mask<<<gS, 256, 0, ctx->stream>>>(d_pairs, F, S, 0, 1, RF);
            GD_KERNEL_CHECK();
            if ((rc = gd_fft_r2c_2d(ctx, S, S, B, RF, ZM))) return rc;
            if ((rc = mask_conv(ZM, ZW, A.a00))) return rc;
        }
        if (bco == 1) {
            A.a10 = arr + 1 * B * FF, A.a01 = arr + 2 * B * FF, A.a20 = arr + 3 * B * FF, A.a02 = arr + 4 * B * FF,
            A.a11 = arr + 5 * B * FF, A.xP = arr + 6 * B * FF, A.yP = arr + 7 * B * FF;
            struct Mom {
                int px, py;
                double* mask_dst;
                double* hist_dst;
            } moms[5] = {{1, 0, A.a10, A.xP}, {0, 1, A.a01, A.yP}, {2, 0, A.a20, nullptr}, {0, 2, A.a02, nullptr},
                         {1, 1, A.a11, nullptr}};
            for (const Mom& m : moms) {
                if (ov) {
                    if ((rc = direct_moment(ov->d_mask_bc, m.px, m.py, m.mask_dst))) return rc;
                } else {
                    k_fill_window<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_wsum, S, m.px, m.py, RF);
                    GD_KERNEL_CHECK();
                    if ((rc = gd_fft_r2c_2d(ctx, S, S, B, RF, ZK))) return rc;
                    if ((rc = mask_conv(ZM, ZK, m.mask_dst))) return rc;
                }
                if (m.hist_dst) {
                    k_fill_window_rect<<<gN, 256, 0, ctx->stream>>>(d_pairs, d_wsum, Ny, Nx, m.px, m.py, RC);
                    GD_KERNEL_CHECK();
                    if ((rc = gd_fft_r2c_2d(ctx, Ny, Nx, B, RC, ZKc))) return rc;
                    if ((rc = circ_conv(ZHc, ZKc, m.hist_dst))) return rc;
                }
            }
        }
        k_boundary<false><<<gF, 256, 0, ctx->stream>>>(d_pairs, A, d_mx, (int)FF, bco, nullptr, nullptr, 0, 0, F);
        GD_KERNEL_CHECK();
    }
    if (do_mbc) {
        if (ov) {
            if ((rc = direct_moment(ov->d_mask_mbc, 0, 0, d_a00m))) return rc;
        } else {
            k_fill_mask<<<gS, 256, 0, ctx->stream>>>(d_pairs, F, S, 1, do_bc ? 1 : 0, RF);
            GD_KERNEL_CHECK();
            if ((rc = gd_fft_r2c_2d(ctx, S, S, B, RF, ZM))) return rc;
            if ((rc = mask_conv(ZM, ZW, d_a00m))) return rc;
        }
        for (int round = 0; round < mbc; ++round) {
            k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_P, (int)FF, d_mx);
            GD_KERNEL_CHECK();
            k_box<<<gF, 256, 0, ctx->stream>>>(d_hist, d_P, d_mx, (int)FF, d_box);
            GD_KERNEL_CHECK();
            k_fill_circ<<<gN, 256, 0, ctx->stream>>>(d_box, F, Ny, Nx, RC);
            GD_KERNEL_CHECK();
            if ((rc = gd_fft_r2c_2d(ctx, Ny, Nx, B, RC, ZKc))) return rc;
            if ((rc = circ_conv(ZKc, ZWc, d_conv))) return rc;
            if (ov && ov->d_zero)
                k_mbc_update_masked<<<2048, 256, 0, ctx->stream>>>(d_P, d_conv, d_a00m, ov->d_zero, B * FF);
            else
                k_mbc_update<<<2048, 256, 0, ctx->stream>>>(d_P, d_conv, d_a00m, B * FF);
            GD_KERNEL_CHECK();
        }
    }
    if (ov && ov->d_zero) {  // bins2D[bool_mask] = 0  (mcsamples.py:1978-1979)
        k_zero_masked<<<2048, 256, 0, ctx->stream>>>(d_P, ov->d_zero, B * FF);
        GD_KERNEL_CHECK();
    }
    k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_P, (int)FF, d_mx);
    GD_KERNEL_CHECK();
    k_normalise<<<gF, 256, 0, ctx->stream>>>(d_P, d_mx, (int)FF, d_status);
    GD_KERNEL_CHECK();
    if (wait) {
        GD_TRY(gd_fetch(ctx, status_out, d_status, (size_t)B * 4));
        GD_TRY(gd_stream_sync(ctx));
    } else {
        GD_TRY(gd_fetch_pinned(ctx, status_out, d_status, (size_t)B * 4));  // page-locked by the entry point's contract
    }
    return GD_OK;
}

extern "C" {

static int density2d_main(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const double* rx, const double* ry,
                          const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc,
                          void* d_P_out, int32_t* status_out, const MaskOv* ov, bool wait = true,
                          const int32_t* hist_index = nullptr) {
    GD_REQUIRE(ctx && d_hist_v && rx && ry && corr && winw && flags && d_P_out && status_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096, "fine_bins_2D out of range");
    GD_REQUIRE(bco >= -1 && bco <= 1, "unknown boundary_correction_order (expected 0 or 1)");
    GD_REQUIRE(mbc >= 0 && mbc <= 8, "mult_bias_correction_order out of range");
    const double* d_hist = (const double*)d_hist_v;
    double* d_P = (double*)d_P_out;
    std::vector<D2Pair> hp((size_t)B);
    int maxw = 1;
    bool any_limits = false;
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(winw[b] >= 1 && winw[b] <= 2 * F, "bad window half-width");
        GD_REQUIRE(rx[b] > 0 && ry[b] > 0 && fabs(corr[b]) < 1, "bad bandwidth matrix");
        const double a = ry[b] * ry[b], d = rx[b] * rx[b], o = rx[b] * ry[b] * corr[b];
        const double det = a * d - o * o;
        hp[b].c00 = d / det;
        hp[b].c11 = a / det;
        hp[b].c10 = -o / det;
        hp[b].w = winw[b];
        hp[b].hidx = hist_index ? hist_index[b] : b;
        hp[b].pad = 0;
        GD_REQUIRE(hp[b].hidx >= 0, "negative histogram index");
        hp[b].flags = flags[b] & 127;
        // bit 6 = "has_prior" (mcsamples.py:1794): default it from the limit bits for callers that do not set it
        if ((hp[b].flags & 15) != 0) hp[b].flags |= 64;
        if (winw[b] > maxw) maxw = winw[b];
        if (hp[b].flags & 64) any_limits = true;
        GD_REQUIRE((hp[b].flags & 48) == (flags[0] & 48), "a batch must not mix periodic and non-periodic pairs");
    }
    const int per = flags[0] & 48;
    if (per) {
        if (hist_index) {  // the periodic route reads its batch contiguously: gather first (rare)
            GD_REQUIRE(!ov, "an explicit prior mask comes with its pair's histogram, not with an index list");
            double* d_sub = (double*)gd_scratch2(ctx, (int64_t)B * F * F * 8);
            if (!d_sub) return GD_ERR_NOMEM;
            GD_TRY(gd_gather_items(ctx, d_sub, d_hist, hist_index, B, (int64_t)F * F * 8));
            return density2d_periodic(ctx, B, F, d_sub, hp, maxw, per, bco, mbc, d_P, status_out, wait);
        }
        return density2d_periodic(ctx, B, F, d_hist, hp, maxw, per, bco, mbc, d_P, status_out, wait, ov);
    }
    const bool do_bc = any_limits && bco >= 0;
    const int S = next_fft_size(F + 2 * maxw);
    const int Sh = S / 2 + 1;
    const int64_t FF = (int64_t)F * F, SS = (int64_t)S * S, SC = (int64_t)S * Sh;
    // prior-mask moments come from summed-area tables of the window (k_window_sat / k_mask_eval)
    const int n_mom = (do_bc ? (bco == 1 ? 6 : 1) : 0) + (mbc ? 1 : 0);
    const int64_t sat_stride = (int64_t)(2 * maxw + 2) * (2 * maxw + 2);
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    // ---- route: transforms in LDS (three kernels per convolution) where the frame, the window table and the grid fit
    FftDev pl, plH;  // plans of the frame length (columns) and of half of it (the real rows, packed)
    const double2* d_tw = nullptr;
    const int Mmax = 2 * maxw + 1;
    const int RPB = 16;  // rows per block of the row passes
    const size_t lds_budget = 150u * 1024u;
    // the row passes keep the twiddle table in LDS where it fits next to the 16 row buffers (not at S = 1152: read in place)
    const bool rows_twg = ((size_t)S + (size_t)RPB * (S / 2 + 1)) * 16 > 156u * 1024u;
    const size_t lds_rows = ((size_t)(rows_twg ? 0 : S) + (size_t)RPB * (S / 2 + 1)) * 16;  // twiddles + a half-length buffer per row (+ 1: bank padding, X[H])
    // columns per block of the column passes: 8 where the LDS has room (every frame up to 512 points), 7 at S = 1152
    const int ncol = (int)std::max<int64_t>(1, std::min<int64_t>(8, ((int64_t)lds_budget - (int64_t)S * 16) / ((int64_t)S * 16)));
    const size_t lds_cols = ((size_t)S + (size_t)ncol * S) * 16;
    // the window spectra: the (2 maxw + 1)^2 window values sit in LDS next to the column buffers while at least four columns
    // fit beside them; a wider window is read from a table in global memory (k_win_table, round 6)
    const size_t wtab_lds = (size_t)(Mmax * Mmax + 1) / 2 * 16;
    const int64_t ncol_win_lds = ((int64_t)lds_budget - (int64_t)S * 16 - (int64_t)wtab_lds) / ((int64_t)S * 16);
    const bool wglob = ncol_win_lds < 4 || getenv("GDHIP_CONV_WIN_GLOBAL") != nullptr;
    const int ncol_win = wglob ? ncol : (int)std::min<int64_t>(8, ncol_win_lds);
    const size_t lds_win = ((size_t)S + (wglob ? 0 : wtab_lds / 16) + (size_t)ncol_win * S) * 16;  // + the window table
    // frames up to 1152 points (round 6: the up-scaled grid classes F = 768 / 960 and the wide windows; until then 512 points
    // and windows whose table fits the LDS beside eight columns -- GDHIP_CONV_LDS_OLD_LIMITS: that rule, for A/B runs)
    const bool size_ok = getenv("GDHIP_CONV_LDS_OLD_LIMITS") ? (S <= 512 && ncol_win_lds >= 8) : S <= 1152;
    const bool lds_conv = !ov && size_ok && (F + RPB - 1) / RPB <= PM_PARTS &&
                          getenv("GDHIP_CONV_ROCFFT") == nullptr && lds_fft_plan(ctx, S, &pl, &d_tw) &&
                          lds_fft_plan(ctx, S / 2, &plH, nullptr);
    if (getenv("GDHIP_CONV_LOG"))  // (route of every batch, for the evidence scripts)
        fprintf(stderr, "gdhip conv: B=%d F=%d maxw=%d S=%d lds_win=%zu cols=%d/%d wglob=%d rows_twg=%d route=%s\n", B, F, maxw, S, lds_win,
                ncol_win, ncol, (int)wglob, (int)rows_twg, lds_conv ? "lds" : "rocfft");
    // the prior-mask moments are evaluated inside their consumers (k_boundary<true>, k_rows_inv<1>) on the LDS route
    const bool fused = lds_conv && !ov && getenv("GDHIP_CONV_MOMENT_ARRAYS") == nullptr;
    // the mask moments of a pixel from their class tables (k_mask_tables) wherever no pixel is clipped on both sides
    const bool class_tables = fused && n_mom > 0 && F >= 4 * maxw + 8 && getenv("GDHIP_CONV_NO_CLASS_TABLES") == nullptr;
    const int64_t tab_stride = (int64_t)(2 * maxw + 3) * (2 * maxw + 3);
    const int64_t XT = (int64_t)B * ((F + 15) / 16) * 16 * Sh * 16;  // tiled half spectra of the LDS route (whole tiles of 16 rows)
    const int64_t WT = (int64_t)B * Sh * S * 8;  // a window moment's spectrum by columns (its real or its imaginary part)
    const int64_t o_pairs = take((int64_t)B * sizeof(D2Pair)), o_wsum = take((int64_t)B * 8), o_mx = take((int64_t)B * 8 * PM_PARTS),
                  o_mx2 = take((int64_t)B * 8 * PM_PARTS),
                  o_status = take((int64_t)B * 4), o_RF = take(lds_conv ? XT : B * SS * 8), o_RO = take(lds_conv ? XT : B * SS * 8),
                  o_ZH = take(lds_conv ? WT : B * SC * 16), o_ZW = take(lds_conv ? WT : B * SC * 16),
                  o_ZK = take(!lds_conv && do_bc && bco == 1 ? B * SC * 16 : 0), o_ZP = take(lds_conv ? 0 : B * SC * 16),
                  o_arr = take((do_bc ? (bco == 1 ? (fused ? 2 : 8) : (fused ? 0 : 1)) : 0) * B * FF * 8),
                  o_a00m = take(mbc && !fused ? B * FF * 8 : 0),
                  o_conv = take(mbc ? B * FF * 8 : 0), o_sat = take((int64_t)n_mom * B * sat_stride * 8),
                  o_tab = take(class_tables ? (int64_t)n_mom * B * tab_stride * 8 : 0),
                  o_wtab = take(lds_conv && wglob ? (int64_t)B * Mmax * Mmax * 8 : 0),
                  o_rowdft = take(lds_conv && wglob ? (int64_t)B * Mmax * Sh * 16 : 0);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    D2Pair* d_pairs = (D2Pair*)(base + o_pairs);
    double* d_wsum = (double*)(base + o_wsum);
    double* d_mx = (double*)(base + o_mx);    // block maxima of the current grid ...
    double* d_mx2 = (double*)(base + o_mx2);  // ... and the set the boundary correction writes while it reads the other
    int* d_status = (int*)(base + o_status);
    double* RF = (double*)(base + o_RF);
    double* RO = (double*)(base + o_RO);
    double2* ZH = (double2*)(base + o_ZH);
    double2* ZW = (double2*)(base + o_ZW);
    double2* ZK = (double2*)(base + o_ZK);
    double2* ZP = (double2*)(base + o_ZP);
    double* arr = (double*)(base + o_arr);
    double* d_a00m = (double*)(base + o_a00m);
    double* d_conv = (double*)(base + o_conv);
    double* d_sat = (double*)(base + o_sat);
    double* d_tab = class_tables ? (double*)(base + o_tab) : nullptr;
    double* d_wtab = (double*)(base + o_wtab);
    double2* d_rowdft = (double2*)(base + o_rowdft);
    if (wait) {
        GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(D2Pair)));
    } else {  // hp dies with this frame before the copy executes
        const int rc_stage = gd_stage_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(D2Pair));
        if (rc_stage) return rc_stage;
    }
    const dim3 gS(128, B), gF(PM_PARTS, B);
    const double scale = 1.0 / ((double)S * (double)S);
    const int cm_blocks = 2048;
    int rc;
#define FWD(frames, Z)                                        \
    do {                                                      \
        rc = gd_fft_r2c_2d(ctx, S, S, B, frames, Z);          \
        if (rc) return rc;                                    \
    } while (0)
    // product of two spectra -> inverse -> crop into dst (B x F x F), the crop leaving its block maxima in `mxp`
    // (nullptr: not wanted)
#define CONV_TO(ZA, ZB, dst, mxp)                                                                        \
    do {                                                                                                 \
        k_cmul<<<cm_blocks, 256, 0, ctx->stream>>>(ZA, ZB, B * SC, scale, ZP);                           \
        GD_KERNEL_CHECK();                                                                               \
        rc = gd_fft_c2r_2d(ctx, S, S, B, ZP, RO);                                                        \
        if (rc) return rc;                                                                               \
        k_crop_fused<0><<<gF, 256, 0, ctx->stream>>>(d_pairs, RO, F, S, dst, nullptr, mxp);              \
        GD_KERNEL_CHECK();                                                                               \
    } while (0)

    k_win_sum<<<B, 256, 0, ctx->stream>>>(d_pairs, d_wsum);
    GD_KERNEL_CHECK();
    // LDS route: Xt = row spectra of the source (RF's block), Yt = columns after the convolution (RO's block)
    double2* Xt = (double2*)RF;
    double2* Yt = (double2*)RO;
    const dim3 gR((F + RPB - 1) / RPB, B), gC((Sh + ncol - 1) / ncol, B), gW((Sh + ncol_win - 1) / ncol_win, B);
    // ---- LDS route: launches.  ZW's block holds the plain window's spectrum by columns (kept for the bias-correction
    //      round), ZH's block the moment windows' one after the other.
    const int sz = S <= 320 ? 0 : S <= 512 ? 1 : 2;  // size class of the transforms (butterflies per lane; 2: 64-lane groups)
    const int ftn = sz == 2 ? 64 : 32;
    // 288-point frames: the rows' 144-point register transform on 16-lane groups (four rows per wave); GDHIP_CONV_ROWS32=1: the
    // 32-lane groups of round 4 (A/B switch)
    const bool rows16 = S == 288 && getenv("GDHIP_CONV_ROWS32") == nullptr;
    double* Wt0 = (double*)ZW;
    double* Wt1 = (double*)ZH;
    // the window moment's spectrum
    auto lds_win_spec = [&](int px_, int py_, double* WT_) -> int {
        if (wglob) {
            k_win_table<<<dim3(8, B), 256, 0, ctx->stream>>>(d_pairs, d_wsum, px_, py_, maxw, d_wtab);
            GD_KERNEL_CHECK();
            k_win_rowdft<<<dim3(Mmax, (Sh + 255) / 256, B), 256, (size_t)S * 16, ctx->stream>>>(d_pairs, d_wtab, S, d_tw, maxw, d_rowdft);
            GD_KERNEL_CHECK();
        }
        auto kern = wglob ? (sz == 2 ? k_win_spec<2, true> : sz == 1 ? k_win_spec<1, true> : k_win_spec<0, true>)
                          : (sz == 2 ? k_win_spec<2, false> : sz == 1 ? k_win_spec<1, false> : k_win_spec<0, false>);
        GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        kern<<<gW, ncol_win * ftn, lds_win, ctx->stream>>>(d_pairs, d_wsum, pl, d_tw, px_, py_, maxw, WT_, d_rowdft);
        GD_KERNEL_CHECK();
        return GD_OK;
    };
    // row spectra of a source into Xt (box: the bias-correction box of src and P)
    auto lds_rows_fwd = [&](bool box, const double* P_, const double* mx_) -> int {
        auto kern = rows16    ? (box ? k_rows_fwd<1, 16> : k_rows_fwd<0, 16>)
                    : sz != 2 ? (box ? k_rows_fwd<1, 32> : k_rows_fwd<0, 32>)
                    : rows_twg ? (box ? k_rows_fwd<1, 64, true> : k_rows_fwd<0, 64, true>)
                               : (box ? k_rows_fwd<1, 64> : k_rows_fwd<0, 64>);
        GD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
        kern<<<gR, RPB * (rows16 ? 16 : ftn), lds_rows, ctx->stream>>>(d_pairs, d_hist, P_, mx_, F, plH, d_tw, Xt);
        GD_KERNEL_CHECK();
        return GD_OK;
    };
    // convolution of the source in Xt with the window spectrum WT_, cropped into dst (update: dst *= crop / a00)
    auto lds_conv_to = [&](const double* WT_, int w_odd, bool update, double* dst, const double* a00_, double* mxp) -> int {
        if ((S == 288 || S == 320 || S == 384) && getenv("GDHIP_CONV_RADIX_PASSES") == nullptr) {  // (the switch: A/B tests)
            const int M = S / 16, cols = 3 * (64 / M);
            const size_t lds16 = (size_t)(1 + cols) * S * 16;
            auto k16 = M == 18 ? k_col_conv16<18> : M == 20 ? k_col_conv16<20> : k_col_conv16<24>;
            GD_HIP(hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
            k16<<<dim3((Sh + cols - 1) / cols, B), 192, lds16, ctx->stream>>>(d_pairs, F, d_tw, WT_, w_odd, Xt, Yt);
        } else {
            auto kc = sz == 2 ? k_col_conv<2> : sz == 1 ? k_col_conv<1> : k_col_conv<0>;
            GD_HIP(hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cols));
            kc<<<gC, ncol * ftn, lds_cols, ctx->stream>>>(d_pairs, F, pl, d_tw, WT_, w_odd, Xt, Yt);
        }
        GD_KERNEL_CHECK();
        auto kr = rows16    ? (update ? k_rows_inv<1, 16> : k_rows_inv<0, 16>)
                  : sz != 2 ? (update ? k_rows_inv<1, 32> : k_rows_inv<0, 32>)
                  : rows_twg ? (update ? k_rows_inv<1, 64, true> : k_rows_inv<0, 64, true>)
                             : (update ? k_rows_inv<1, 64> : k_rows_inv<0, 64>);
        const bool tabs = update && fused && class_tables;  // the divisor by class: one table row per row of the tile
        size_t lds_rows_inv = lds_rows + (tabs ? (size_t)RPB * (2 * maxw + 3) * 8
                                               : (update && fused ? (size_t)(2 * maxw + 2) * (2 * maxw + 2) * 8 : 0));
        const int sat_in_lds = lds_rows_inv <= 156u * 1024u;  // (else the table is read where it lies)
        if (!sat_in_lds) lds_rows_inv = lds_rows;
        GD_HIP(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows_inv));
        // (update && fused: the divisor comes from the all-edge mask's table, the last of the pair's n_mom tables)
        const double* sat1 = (update && fused && !tabs) ? d_sat + (int64_t)(n_mom - 1) * sat_stride : nullptr;
        const double* tab1 = (tabs && sat_in_lds) ? d_tab + (int64_t)(n_mom - 1) * tab_stride : nullptr;
        if (tabs && !tab1) sat1 = d_sat + (int64_t)(n_mom - 1) * sat_stride;  // (no room for the rows: the older path)
        kr<<<gR, RPB * (rows16 ? 16 : ftn), lds_rows_inv, ctx->stream>>>(d_pairs, Yt, F, plH, d_tw, dst, a00_, mxp, sat1, (int64_t)n_mom * sat_stride,
                                                        do_bc ? 1 : 0, sat_in_lds, tab1, (int64_t)n_mom * tab_stride);
        GD_KERNEL_CHECK();
        return GD_OK;
    };
    if (lds_conv) {
        GD_TRY(lds_rows_fwd(false, nullptr, nullptr));
        GD_TRY(lds_win_spec(0, 0, Wt0));
        GD_TRY(lds_conv_to(Wt0, 0, false, d_P, nullptr, d_mx));
    } else {
        // spectra of the window and of the histogram
        k_fill_window<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_wsum, S, 0, 0, RF);
        GD_KERNEL_CHECK();
        FWD(RF, ZW);
        k_fill_embed<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_hist, F, S, RF);
        GD_KERNEL_CHECK();
        FWD(RF, ZH);
        CONV_TO(ZH, ZW, d_P, d_mx);  // bins2D = conv(histbins, Win, 'same')   (mcsamples.py:1884), with its maxima
    }
    BcArrays A;
    A.P = d_P;
    A.a00 = arr;
    A.a10 = A.a01 = A.a20 = A.a02 = A.a11 = A.xP = A.yP = nullptr;
    if (do_bc && bco == 1 && !fused)
        A.a10 = arr + 1 * B * FF, A.a01 = arr + 2 * B * FF, A.a20 = arr + 3 * B * FF, A.a02 = arr + 4 * B * FF,
        A.a11 = arr + 5 * B * FF, A.xP = arr + 6 * B * FF, A.yP = arr + 7 * B * FF;
    if (fused) {  // no moment arrays: only the two histogram-side convolutions of the linear correction are grids
        A.a00 = nullptr;
        if (do_bc && bco == 1) A.xP = arr, A.yP = arr + B * FF;
    }
    if (n_mom) {
        MomList L;
        L.n = 0;
        L.edge_applied = do_bc ? 1 : 0;
        if (do_bc) {
            L.m[L.n++] = {0, 0, 0, A.a00};
            if (bco == 1) {
                L.m[L.n++] = {1, 0, 0, A.a10};
                L.m[L.n++] = {0, 1, 0, A.a01};
                L.m[L.n++] = {2, 0, 0, A.a20};
                L.m[L.n++] = {0, 2, 0, A.a02};
                L.m[L.n++] = {1, 1, 0, A.a11};
            }
        }
        if (mbc) L.m[L.n++] = {0, 0, 1, d_a00m};
        if (!ov) {
            if (class_tables && sat_stride * 8 <= 64 * 1024) {  // tables built with the summed-area table in LDS
                GD_HIP(hipFuncSetAttribute((const void*)k_window_sat_tab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sat_stride * 8)));
                k_window_sat_tab<<<dim3(L.n, B), 256, (size_t)(sat_stride * 8), ctx->stream>>>(d_pairs, d_wsum, L, F, sat_stride, d_sat,
                                                                                               tab_stride, d_tab);
                GD_KERNEL_CHECK();
            } else {
                k_window_sat<<<dim3(L.n, B), 256, 0, ctx->stream>>>(d_pairs, d_wsum, L, sat_stride, d_sat);
                GD_KERNEL_CHECK();
                if (class_tables) {
                    k_mask_tables<<<dim3(L.n, B), 256, 0, ctx->stream>>>(d_pairs, L, F, sat_stride, d_sat, tab_stride, d_tab);
                    GD_KERNEL_CHECK();
                }
            }
            if (!fused) {
                k_mask_eval<<<dim3(32, L.n, B), 256, 0, ctx->stream>>>(d_pairs, L, F, sat_stride, d_sat);
                GD_KERNEL_CHECK();
            }
        } else {
            for (int q = 0; q < L.n; ++q) {  // d_sat doubles as the window-moment buffer ((2w+1)^2 <= sat_stride)
                const double* mask = L.m[q].kind == 0 ? ov->d_mask_bc : ov->d_mask_mbc;
                GD_REQUIRE(mask, "explicit mask missing for a requested correction");
                k_moment_window<<<64, 256, 0, ctx->stream>>>(d_pairs, d_wsum, L.m[q].px, L.m[q].py, d_sat);
                GD_KERNEL_CHECK();
                k_mask_moment_direct<<<1024, 256, 0, ctx->stream>>>(d_sat, mask, F, maxw, L.m[q].dst);
                GD_KERNEL_CHECK();
            }
        }
    }
    double* mx_cur = d_mx;
    if (do_bc) {
        if (bco == 1 && lds_conv) {
            // x*P and y*P still need the histogram: conv(histbins, Win*x), conv(histbins, Win*y)  (mcsamples.py:1940-1941)
            GD_TRY(lds_win_spec(1, 0, Wt1));
            GD_TRY(lds_conv_to(Wt1, 1, false, A.xP, nullptr, nullptr));
            GD_TRY(lds_win_spec(0, 1, Wt1));
            GD_TRY(lds_conv_to(Wt1, 1, false, A.yP, nullptr, nullptr));
        } else if (bco == 1) {
            k_fill_window<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_wsum, S, 1, 0, RF);
            GD_KERNEL_CHECK();
            FWD(RF, ZK);
            CONV_TO(ZH, ZK, A.xP, (double*)nullptr);
            k_fill_window<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_wsum, S, 0, 1, RF);
            GD_KERNEL_CHECK();
            FWD(RF, ZK);
            CONV_TO(ZH, ZK, A.yP, (double*)nullptr);
        }
        if (fused && class_tables) {
            k_boundary<true><<<gF, 256, 0, ctx->stream>>>(d_pairs, A, d_mx, (int)FF, bco, d_mx2, d_sat, sat_stride, n_mom, F, 0, d_tab,
                                                          tab_stride);
        } else if (fused) {
            size_t lds_bc = (size_t)(bco == 1 ? 6 : 1) * sat_stride * 8;
            const int stage = lds_bc <= 64u * 1024u;  // (windows up to 17 bins; wider ones read their tables from global memory)
            if (!stage) lds_bc = 0;
            GD_HIP(hipFuncSetAttribute((const void*)k_boundary<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bc));
            k_boundary<true><<<gF, 256, lds_bc, ctx->stream>>>(d_pairs, A, d_mx, (int)FF, bco, d_mx2, d_sat, sat_stride, n_mom, F, stage);
        } else {
            k_boundary<false><<<gF, 256, 0, ctx->stream>>>(d_pairs, A, d_mx, (int)FF, bco, d_mx2, nullptr, 0, 0, F);
        }
        GD_KERNEL_CHECK();
        mx_cur = d_mx2;
    }
    if (mbc > 0) {
        for (int round = 0; round < mbc; ++round) {
            if (lds_conv) {
                // bins2D *= conv(box, Win) / a00: the box is formed in the row pass, the update in the inverse row pass
                GD_TRY(lds_rows_fwd(true, d_P, mx_cur));
                GD_TRY(lds_conv_to(Wt0, 0, true, d_P, d_a00m, mx_cur));
                continue;
            }
            k_fill_box<<<gS, 256, 0, ctx->stream>>>(d_pairs, d_hist, d_P, mx_cur, F, S, RF);
            GD_KERNEL_CHECK();
            FWD(RF, ZH);  // ZH is free to reuse: the histogram spectrum is no longer needed
            if (ov && ov->d_zero) {
                CONV_TO(ZH, ZW, d_conv, (double*)nullptr);
                k_mbc_update_masked<<<cm_blocks, 256, 0, ctx->stream>>>(d_P, d_conv, d_a00m, ov->d_zero, B * FF);
                GD_KERNEL_CHECK();
                k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_P, (int)FF, mx_cur);
                GD_KERNEL_CHECK();
            } else {
                // bins2D *= conv(box, Win) / a00 straight out of the inverse transform's frame, maxima included
                k_cmul<<<cm_blocks, 256, 0, ctx->stream>>>(ZH, ZW, B * SC, scale, ZP);
                GD_KERNEL_CHECK();
                rc = gd_fft_c2r_2d(ctx, S, S, B, ZP, RO);
                if (rc) return rc;
                k_crop_fused<1><<<gF, 256, 0, ctx->stream>>>(d_pairs, RO, F, S, d_P, d_a00m, mx_cur);
                GD_KERNEL_CHECK();
            }
        }
    }
    if (ov && ov->d_zero) {  // bins2D[bool_mask] = 0  (mcsamples.py:1978-1979)
        k_zero_masked<<<cm_blocks, 256, 0, ctx->stream>>>(d_P, ov->d_zero, B * FF);
        GD_KERNEL_CHECK();
        k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_P, (int)FF, mx_cur);
        GD_KERNEL_CHECK();
    }
    k_normalise<<<gF, 256, 0, ctx->stream>>>(d_P, mx_cur, (int)FF, d_status);
    GD_KERNEL_CHECK();
    if (wait) {
        GD_TRY(gd_fetch(ctx, status_out, d_status, (size_t)B * 4));
        GD_TRY(gd_stream_sync(ctx));
    } else {
        GD_TRY(gd_fetch_pinned(ctx, status_out, d_status, (size_t)B * 4));  // page-locked by the entry point's contract
    }
#undef FWD
#undef CONV_TO

    return GD_OK;
}

int gd_density2d(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const double* rx, const double* ry,
                 const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc, void* d_P_out,
                 int32_t* status_out) {
    return density2d_main(ctx, B, F, d_hist_v, rx, ry, corr, winw, flags, bco, mbc, d_P_out, status_out, nullptr);
}

int gd_density2d_enqueue(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const double* rx, const double* ry,
                         const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc,
                         void* d_P_out, int32_t* status_pinned) {
    return density2d_main(ctx, B, F, d_hist_v, rx, ry, corr, winw, flags, bco, mbc, d_P_out, status_pinned, nullptr, false);
}

int gd_density2d_enqueue_indexed(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const int32_t* hist_index,
                                 const double* rx, const double* ry, const double* corr, const int32_t* winw, const int32_t* flags,
                                 int32_t bco, int32_t mbc, void* d_P_out, int32_t* status_pinned) {
    return density2d_main(ctx, B, F, d_hist_v, rx, ry, corr, winw, flags, bco, mbc, d_P_out, status_pinned, nullptr, false, hist_index);
}

int gd_density2d_masked(gd_ctx* ctx, int32_t F, const void* d_hist, double rx, double ry, double corr, int32_t winw,
                        int32_t flags, int32_t bco, int32_t mbc, const double* mask_bc, const double* mask_mbc,
                        const unsigned char* zero_mask, void* d_P_out, int32_t* status_out) {
    GD_REQUIRE(ctx && d_hist && d_P_out && status_out, "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096 && winw >= 1 && winw <= 2 * F, "bad grid size / window half-width");
    const int64_t Mp = (int64_t)F + 2 * winw, nb = (Mp * Mp * 8 + 255) / 256 * 256, nz = ((int64_t)F * F + 255) / 256 * 256;
    char* base = (char*)gd_scratch2(ctx, 2 * nb + nz);
    if (!base) return GD_ERR_NOMEM;
    MaskOv ov{nullptr, nullptr, nullptr};
    if (mask_bc) {
        GD_TRY(gd_h2d(ctx, base, mask_bc, (size_t)(Mp * Mp * 8)));
        ov.d_mask_bc = (const double*)base;
    }
    if (mask_mbc) {
        GD_TRY(gd_h2d(ctx, base + nb, mask_mbc, (size_t)(Mp * Mp * 8)));
        ov.d_mask_mbc = (const double*)(base + nb);
    }
    if (zero_mask) {
        GD_TRY(gd_h2d(ctx, base + 2 * nb, zero_mask, (size_t)F * F));
        ov.d_zero = (const unsigned char*)(base + 2 * nb);
    }
    const int32_t fl = flags | 64;  // a mask function always counts as a prior (mcsamples.py:1794)
    return density2d_main(ctx, 1, F, d_hist, &rx, &ry, &corr, &winw, &fl, bco, mbc, d_P_out, status_out, &ov);
}

int gd_likes2d(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist_v, const void* d_likehist_v, const double* rx,
               const double* ry, const double* corr, const int32_t* winw, const int32_t* flags, int32_t mbc,
               void* d_likes_out, int32_t* status_out) {
    GD_REQUIRE(ctx && d_hist_v && d_likehist_v && rx && ry && corr && winw && flags && d_likes_out && status_out && B > 0,
               "bad argument");
    GD_REQUIRE(F >= 8 && F <= 4096, "fine_bins_2D out of range");
    const double* d_hist = (const double*)d_hist_v;
    const double* d_likehist = (const double*)d_likehist_v;
    double* d_L = (double*)d_likes_out;
    std::vector<D2Pair> hp((size_t)B);
    int maxw = 1;
    for (int b = 0; b < B; ++b) {
        GD_REQUIRE(winw[b] >= 1 && winw[b] <= 2 * F, "bad window half-width");
        GD_REQUIRE(rx[b] > 0 && ry[b] > 0 && fabs(corr[b]) < 1, "bad bandwidth matrix");
        const double a = ry[b] * ry[b], d = rx[b] * rx[b], o = rx[b] * ry[b] * corr[b];
        const double det = a * d - o * o;
        hp[b].c00 = d / det;
        hp[b].c11 = a / det;
        hp[b].c10 = -o / det;
        hp[b].w = winw[b];
        hp[b].hidx = b;
        hp[b].pad = 0;
        hp[b].flags = flags[b] & 127;
        if (winw[b] > maxw) maxw = winw[b];
        GD_REQUIRE((hp[b].flags & 48) == (flags[0] & 48), "a batch must not mix periodic and non-periodic pairs");
    }
    GD_REQUIRE(2 * maxw + 1 <= CONV_WIN_DOUBLES, "window too wide for the direct convolution");
    const int per = flags[0] & 48;
    const bool px = per & 16, py = per & 32;
    // operands: the F x F grid itself for 'same', the folded circular (n0 x n1) grid for periodic axes
    const int n0 = py ? F - 1 : F, n1 = px ? F - 1 : F;
    if (per) GD_REQUIRE(2 * maxw + 1 <= n1 && 2 * maxw + 1 <= n0, "window wider than the periodic grid");
    const int64_t FF = (int64_t)F * F, NN = (int64_t)n0 * n1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_pairs = take((int64_t)B * sizeof(D2Pair)), o_wsum = take((int64_t)B * 8), o_mx = take((int64_t)B * 8 * PM_PARTS),
                  o_status = take((int64_t)B * 4), o_C1 = take(per ? B * NN * 8 : 0), o_C2 = take(per ? B * NN * 8 : 0),
                  o_P0 = take(B * FF * 8), o_T = take(B * FF * 8), o_L2 = take(B * FF * 8);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    D2Pair* d_pairs = (D2Pair*)(base + o_pairs);
    double* d_wsum = (double*)(base + o_wsum);
    double* d_mx = (double*)(base + o_mx);
    int* d_status = (int*)(base + o_status);
    double *C1 = (double*)(base + o_C1), *C2 = (double*)(base + o_C2), *d_P0 = (double*)(base + o_P0),
           *d_T = (double*)(base + o_T), *d_L2 = (double*)(base + o_L2);
    GD_TRY(gd_h2d(ctx, d_pairs, hp.data(), (size_t)B * sizeof(D2Pair)));
    const dim3 gN(64, B), gF(64, B);
    const dim3 gC((unsigned)(((n1 + 63) / 64) * ((n0 + 4 * CONV_RY - 1) / (4 * CONV_RY))), B);
    k_win_sum<<<B, 256, 0, ctx->stream>>>(d_pairs, d_wsum);
    GD_KERNEL_CHECK();
    // dst = convolve2D(src, Win, convolution_mode), by direct summation
    auto smooth = [&](const double* src, double* dst) -> int {
        if (per) {
            k_fill_circ<<<gN, 256, 0, ctx->stream>>>(src, F, n0, n1, C1);
            k_conv2d_direct<<<gC, 256, 0, ctx->stream>>>(d_pairs, d_wsum, C1, n0, n1, 1, C2);
            k_expand_circ<<<gF, 256, 0, ctx->stream>>>(C2, F, n0, n1, dst);
        } else {
            k_conv2d_direct<<<gC, 256, 0, ctx->stream>>>(d_pairs, d_wsum, src, n0, n1, 0, dst);
        }
        if (hipGetLastError() != hipSuccess) return gd_fail(ctx, GD_ERR_HIP, "convolution launch failed");
        return GD_OK;
    };
    int rc;
    if ((rc = smooth(d_hist, d_P0))) return rc;       // bins2D before any correction (mcsamples.py:1884)
    if ((rc = smooth(d_likehist, d_L))) return rc;    // bin2Dlikes (:1887)
    if (mbc) {                                        // :1890-1897
        k_likes_div<<<2048, 256, 0, ctx->stream>>>(d_likehist, d_L, B * FF, d_T);
        GD_KERNEL_CHECK();
        if ((rc = smooth(d_T, d_L2))) return rc;
        k_likes_mul<<<2048, 256, 0, ctx->stream>>>(d_L2, B * FF, d_L);
        GD_KERNEL_CHECK();
    }
    k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_P0, (int)FF, d_mx);
    GD_KERNEL_CHECK();
    k_likes_ratio<<<gF, 256, 0, ctx->stream>>>(d_P0, d_mx, (int)FF, d_L);  // :1899-1901
    GD_KERNEL_CHECK();
    k_pair_max<<<dim3(PM_PARTS, B), 256, 0, ctx->stream>>>(d_L, (int)FF, d_mx);
    GD_KERNEL_CHECK();
    k_normalise<<<gF, 256, 0, ctx->stream>>>(d_L, d_mx, (int)FF, d_status);  // bin2Dlikes /= max (:2005)
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, status_out, d_status, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

}  // extern "C"
