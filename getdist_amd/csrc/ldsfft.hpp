// ldsfft.hpp -- Stockham transforms in LDS (device code) and their plans (host), shared by the convolution
// (density2d.hip) and the optimiser's DCT (kopt2d.hip).  Not part of the ABI.
#pragma once
#include "ctx.hpp"

#define FT 32
struct FftDev {
    int S, nst;
    int radix[12];
    unsigned int magic[12];  // ceil(2^20 / Ns) of pass st (Ns = product of the earlier radices): j / Ns = (j * magic) >> 20
                             // exactly for j < 1024 and Ns <= 1024 (the excess e = magic Ns - 2^20 < Ns, and the quotient is
                             // exact while j e < 2^20) -- the passes are VALU-bound and a runtime division costs ~40 instructions
};

// The FT lanes of a transform sit inside one wavefront, whose LDS operations execute in order: passes are separated by a
// compiler-level fence, not by a block barrier, so the waves of a block run their transforms independently.
__device__ __forceinline__ void group_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// complex product with explicit fused multiply-adds (the library is built with -ffp-contract=off for the kernels whose
// results must match numpy's operation by operation; these transforms are VALU-bound and have no such twin)
__device__ __forceinline__ double2 cmulf(const double2 a, const double2 b) {
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}

// One in-place Stockham pass of radix R over the length-S sequence in `buf`; MAXIT >= ceil(S / R / FT) butterflies per
// lane; INV: inverse transform (conjugate twiddles).  `emit(pos, value)` receives the outputs after the group has read all
// its inputs (default: store to buf).  tw[k * tws] = e^{-2 pi i k / S}: a table made for a multiple of S serves with its
// stride.  Groups without work run the passes on their (unused) buffer like the others -- the passes are VALU-bound and a
// per-lane `active` test in every butterfly costs more than the idle arithmetic of the last block of a grid.
template <int R, int MAXIT, bool INV, int FTN = FT, class Emit>
__device__ __forceinline__ void fft_pass(double2* __restrict__ buf, const double2* __restrict__ tw, int tws, int S, int Ns,
                                         unsigned int magic, int nb, int t, Emit emit) {
    const int step = (nb / Ns) * tws;  // S / (Ns R) twiddle-table entries per unit of q k
    const bool twiddles = Ns > 1;      // (wave-uniform) the first pass has none
    double2 o[MAXIT][R];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int j = t + it * FTN;
        if (j < nb) {
            const int k = j - (int)(((unsigned int)j * magic) >> 20) * Ns;  // j % Ns
            double2 v[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                double2 x = buf[j + q * nb];
                if (q > 0 && twiddles) {
                    double2 w = tw[q * k * step];
                    if (INV) w.y = -w.y;
                    x = cmulf(x, w);
                }
                v[q] = x;
            }
            if (R == 2) {
                o[it][0] = make_double2(v[0].x + v[1].x, v[0].y + v[1].y);
                o[it][1] = make_double2(v[0].x - v[1].x, v[0].y - v[1].y);
            } else if (R == 4) {
                const double2 t0 = make_double2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_double2(v[0].x - v[2].x, v[0].y - v[2].y);
                const double2 t2 = make_double2(v[1].x + v[3].x, v[1].y + v[3].y);
                const double2 d = make_double2(v[1].x - v[3].x, v[1].y - v[3].y);
                const double2 t3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);  // d * (+i) : d * (-i)
                o[it][0] = make_double2(t0.x + t2.x, t0.y + t2.y);
                o[it][1] = make_double2(t1.x + t3.x, t1.y + t3.y);
                o[it][2] = make_double2(t0.x - t2.x, t0.y - t2.y);
                o[it][3] = make_double2(t1.x - t3.x, t1.y - t3.y);
            } else if (R == 3) {
                // o0 = v0 + (v1 + v2);  o1, o2 = v0 - (v1 + v2) / 2  -+ i sin(60) (v1 - v2)   (forward; + - for the inverse)
                constexpr double SIN60 = 0.86602540378443864676;
                const double2 sm = make_double2(v[1].x + v[2].x, v[1].y + v[2].y), df = make_double2(v[1].x - v[2].x, v[1].y - v[2].y);
                const double2 md = make_double2(fma(-0.5, sm.x, v[0].x), fma(-0.5, sm.y, v[0].y));
                // -i s df = (s df.y, -s df.x) forward;  +i s df = (-s df.y, s df.x) inverse
                const double rx = (INV ? -SIN60 : SIN60) * df.y, ry = (INV ? SIN60 : -SIN60) * df.x;
                o[it][0] = make_double2(v[0].x + sm.x, v[0].y + sm.y);
                o[it][1] = make_double2(md.x + rx, md.y + ry);
                o[it][2] = make_double2(md.x - rx, md.y - ry);
            } else {  // small DFT by its definition; cos / sin of 2 pi m / R as literals (R = 5)
                constexpr double C3[3] = {1.0, -0.5, -0.5};
                constexpr double S3[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
                constexpr double C5[5] = {1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410,
                                          0.30901699437494742410};
                constexpr double S5[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917,
                                          -0.95105651629515357212};
#pragma unroll
                for (int p = 0; p < R; ++p) {
                    double2 acc = v[0];
#pragma unroll
                    for (int q = 1; q < R; ++q) {
                        const int m = (p * q) % R;
                        const double c = R == 3 ? C3[m] : C5[m];
                        const double sn = (R == 3 ? S3[m] : S5[m]) * (INV ? 1.0 : -1.0);  // e^{-+ 2 pi i m / R}
                        acc.x = fma(v[q].x, c, fma(-v[q].y, sn, acc.x));
                        acc.y = fma(v[q].x, sn, fma(v[q].y, c, acc.y));
                    }
                    o[it][p] = acc;
                }
            }
        }
    }
    group_sync();
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int j = t + it * FTN;
        if (j < nb) {
            const int jq = (int)(((unsigned int)j * magic) >> 20);  // j / Ns
            const int j0 = jq * Ns * (R - 1) + j;                   // (j / Ns) Ns R + j % Ns
#pragma unroll
            for (int q = 0; q < R; ++q) emit(j0 + q * Ns, o[it][q]);
        }
    }
    group_sync();
}

// butterflies per lane: ceil(S / R / lanes).  Size class SZ = 0 for S <= 320 (the triangle's frames), 1 up to 512 -- both on
// FTN = 32 lanes --, 2 up to 1280 on FTN = 64 lanes (the up-scaled grid classes; round 6).  The classes 0 / 1 on 64 lanes
// serve twice their lengths (the half-length row transforms of the large frames).
template <int SZ, bool INV, int FTN = FT, class Emit>
__device__ __forceinline__ void fft_pass_any(const FftDev& pl, int st, double2* buf, const double2* tw, int tws, int Ns, int t,
                                             Emit emit) {
    const int R = pl.radix[st], S = pl.S;
    const unsigned int mg = pl.magic[st];
    if (R == 4) fft_pass<4, SZ == 2 ? 5 : SZ == 1 ? 4 : 3, INV, FTN>(buf, tw, tws, S, Ns, mg, S >> 2, t, emit);
    else if (R == 2) fft_pass<2, SZ == 2 ? 10 : SZ == 1 ? 8 : 5, INV, FTN>(buf, tw, tws, S, Ns, mg, S >> 1, t, emit);
    else if (R == 3) fft_pass<3, SZ == 2 ? 7 : SZ == 1 ? 6 : 4, INV, FTN>(buf, tw, tws, S, Ns, mg, S / 3, t, emit);
    else fft_pass<5, SZ == 2 ? 4 : SZ == 1 ? 4 : 2, INV, FTN>(buf, tw, tws, S, Ns, mg, S / 5, t, emit);
}

// all passes but the last; returns the sub-transform length the last pass starts from
template <int SZ, bool INV, int FTN = FT>
__device__ __forceinline__ int fft_head(double2* buf, const double2* tw, int tws, const FftDev& pl, int t) {
    int Ns = 1;
    for (int st = 0; st + 1 < pl.nst; ++st) {
        fft_pass_any<SZ, INV, FTN>(pl, st, buf, tw, tws, Ns, t, [&](int pos, double2 v) { buf[pos] = v; });
        Ns *= pl.radix[st];
    }
    return Ns;
}

template <int SZ, bool INV, int FTN = FT>
__device__ __forceinline__ void fft_full(double2* buf, const double2* tw, int tws, const FftDev& pl, int t) {
    const int Ns = fft_head<SZ, INV, FTN>(buf, tw, tws, pl, t);
    fft_pass_any<SZ, INV, FTN>(pl, pl.nst - 1, buf, tw, tws, Ns, t, [&](int pos, double2 v) { buf[pos] = v; });
}

// The LDS route's plan for frame size S (radices 4, 2, 3, 5) and its twiddle table on the device (cached per context).
// false: S has another prime factor (not on the ladder) -- the caller keeps the rocFFT route.
static inline bool lds_fft_plan(gd_ctx* ctx, int S, FftDev* pl, const double2** tw) {
    pl->S = S, pl->nst = 0;
    int n = S;
    // odd radices first: a pass writes its outputs R * (length of the finished sub-transforms) apart, and a power-of-two
    // stride in the first pass (sub-transform length 1) would put every lane's 16-byte store on a few LDS banks
    const int order[4] = {3, 5, 4, 2};
    for (int q = 0; q < 4; ++q)
        while (n % order[q] == 0 && pl->nst < 12) pl->radix[pl->nst++] = order[q], n /= order[q];
    if (n != 1 || pl->nst == 0) return false;
    for (int st = 0, Ns = 1; st < pl->nst; Ns *= pl->radix[st], ++st) pl->magic[st] = (unsigned int)(((1u << 20) + Ns - 1) / Ns);
    if (!tw) return true;  // plan only (the rows' half-length transforms use the frame's table with stride 2)
    auto it = ctx->fft_tw.find(S);
    if (it == ctx->fft_tw.end()) {
        std::vector<double2> h((size_t)S);
        for (int k = 0; k < S; ++k) {
            const long double a = -2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)S;
            h[k] = make_double2((double)cosl(a), (double)sinl(a));
        }
        void* d = nullptr;
        if (hipMalloc(&d, (size_t)S * 16) != hipSuccess) return false;
        if (hipMemcpy(d, h.data(), (size_t)S * 16, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return false;
        }
        it = ctx->fft_tw.emplace(S, d).first;
    }
    *tw = (const double2*)it->second;
    return true;
}

