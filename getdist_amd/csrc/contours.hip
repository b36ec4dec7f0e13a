// Contour levels of a batch of 2D density grids (densities.py:19-56 getContourLevels with half_edge=True): the
// density value below which a fraction (1 - contour) of the half-edge-weighted grid mass lies, linearly
// interpolated between the two grid values that bracket the crossing.  The reference argsorts the F^2 grid on the
// host (about 4 ms per 256^2 grid, i.e. seconds for a triangle); here one block per grid runs an MSB radix select
// on the (L2-resident) grid: 8 passes of 8 key bits over the un-halved values with the half-edge weights as the
// mass, then one pass for the rows tied with the selected value and for its predecessor in sorted order.
#include "ctx.hpp"

#define CL_MAXC 8
#define CL_TIES 1024

__device__ __forceinline__ unsigned long long cl_key(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double cl_edge_weight(int e, int F) {
    const int r = e / F, c = e % F;
    return ((r == 0 || r == F - 1) ? 0.5 : 1.0) * ((c == 0 || c == F - 1) ? 0.5 : 1.0);
}

__global__ void __launch_bounds__(1024) k_contour_levels(const double* __restrict__ Pall, int F, const double* __restrict__ contours,
                                                         int nc, double* __restrict__ out, int* __restrict__ status) {
    __shared__ double hist[256];
    __shared__ double red[16];
    __shared__ unsigned long long s_prefix;
    __shared__ double s_cum, s_norm;
    __shared__ int tie_idx[CL_TIES];
    __shared__ int n_ties;
    __shared__ unsigned long long s_predkey[16];
    __shared__ int s_predidx[16];
    const int FF = F * F, tid = threadIdx.x;
    const double* P = Pall + (int64_t)blockIdx.x * FF;
    // norm = sum of the half-edge-weighted grid
    double s = 0;
    for (int e = tid; e < FF; e += blockDim.x) s += P[e] * cl_edge_weight(e, F);
    s = block_sum(s, red);
    if (tid == 0) s_norm = s;
    __syncthreads();
    const double norm = s_norm;
    int st = GD_OK;
    for (int ci = 0; ci < nc; ++ci) {
        const double target = (1.0 - contours[ci]) * norm;
        if (tid == 0) s_prefix = 0ull, s_cum = 0.0;
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0.0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            for (int e = tid; e < FF; e += blockDim.x) {
                const double v = P[e];
                const unsigned long long key = cl_key(v);
                if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], v * cl_edge_weight(e, F));
            }
            __syncthreads();
            if (tid == 0) {
                double cum = s_cum;
                int pick = -1, last = -1;
                double cum_last = cum;
                for (int b = 0; b < 256; ++b) {
                    const double hv = hist[b];
                    if (hv != 0) {
                        last = b;
                        cum_last = cum;
                        if (cum + hv >= target) {
                            pick = b;
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
        // rows tied with the selected value (index order) and the largest value below it
        const unsigned long long sel = s_prefix;
        if (tid == 0) n_ties = 0;
        __syncthreads();
        unsigned long long pk = 0ull;
        int pi = -1;
        for (int e = tid; e < FF; e += blockDim.x) {
            const unsigned long long key = cl_key(P[e]);
            if (key == sel) {
                const int pos = atomicAdd(&n_ties, 1);
                if (pos < CL_TIES) tie_idx[pos] = e;
            } else if (key < sel && (pi < 0 || key > pk || (key == pk && e > pi))) {
                pk = key, pi = e;
            }
        }
        // block arg-max of (pk, pi)
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ok = __shfl_down(pk, o, WAVE);
            const int oi = __shfl_down(pi, o, WAVE);
            if (oi >= 0 && (pi < 0 || ok > pk || (ok == pk && oi > pi))) pk = ok, pi = oi;
        }
        if ((tid & 63) == 0) s_predkey[tid >> 6] = pk, s_predidx[tid >> 6] = pi;
        __syncthreads();
        if (tid == 0) {
            for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) {
                const unsigned long long ok = s_predkey[wv];
                const int oi = s_predidx[wv];
                if (oi >= 0 && (pi < 0 || ok > pk || (ok == pk && oi > pi))) pk = ok, pi = oi;
            }
            int nt = n_ties;
            double level = 0.0;
            if (nt > CL_TIES) {
                st = GD_ERR_SOLVER;  // too many exactly equal values at the level: the caller falls back to the host
            } else {
                for (int a = 1; a < nt; ++a) {  // insertion sort by index (nt is almost always 1)
                    const int v = tie_idx[a];
                    int b = a - 1;
                    while (b >= 0 && tie_idx[b] > v) tie_idx[b + 1] = tie_idx[b], --b;
                    tie_idx[b + 1] = v;
                }
                double cum = s_cum;  // mass strictly below the selected value
                int at = -1;
                for (int a = 0; a < nt; ++a) {
                    const double wv = P[tie_idx[a]] * cl_edge_weight(tie_idx[a], F);
                    if (wv != 0 && cum + wv >= target) {
                        at = a;
                        cum += wv;
                        break;
                    }
                    cum += wv;
                }
                if (at < 0) at = nt - 1;  // target beyond the total mass: last element
                const int prev = (at > 0) ? tie_idx[at - 1] : pi;
                if (prev < 0) {
                    st = GD_ERR_EMPTY;  // ix == 0: "Contour level outside plotted ranges"
                } else {
                    const double h = P[tie_idx[at]] * cl_edge_weight(tie_idx[at], F);
                    const double hp = P[prev] * cl_edge_weight(prev, F);
                    const double d = (cum - target) / h;
                    level = h * (1 - d) + d * hp;
                }
            }
            out[(int64_t)blockIdx.x * nc + ci] = level;
        }
        __syncthreads();
    }
    if (tid == 0) status[blockIdx.x] = st;
}

extern "C" {

int gd_contour_levels(gd_ctx* ctx, int32_t B, int32_t F, const void* d_P, const double* contours, int32_t nc, double* out,
                      int32_t* status_out) {
    GD_REQUIRE(ctx && d_P && contours && out && status_out && B > 0, "bad argument");
    GD_REQUIRE(F >= 2 && F <= 4096, "grid size out of range");
    GD_REQUIRE(nc >= 1 && nc <= CL_MAXC, "1..8 contours per call");
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const int64_t o_c = take(nc * 8), o_out = take((int64_t)B * nc * 8), o_st = take((int64_t)B * 4);
    char* base = (char*)gd_scratch(ctx, off);
    if (!base) return GD_ERR_NOMEM;
    GD_TRY(gd_h2d(ctx, base + o_c, contours, (size_t)nc * 8));
    k_contour_levels<<<B, 1024, 0, ctx->stream>>>((const double*)d_P, F, (const double*)(base + o_c), nc,
                                                  (double*)(base + o_out), (int*)(base + o_st));
    GD_KERNEL_CHECK();
    GD_TRY(gd_fetch(ctx, out, base + o_out, (size_t)B * nc * 8));
    GD_TRY(gd_fetch(ctx, status_out, base + o_st, (size_t)B * 4));
    GD_TRY(gd_stream_sync(ctx));
    return GD_OK;
}

}  // extern "C"
