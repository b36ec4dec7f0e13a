// fft288.hpp -- a 288-point complex transform as 16 x 18 with both factors done IN REGISTERS (device code; the codelets
// compile on the host too, where tests/native checks them against the definition).  Not part of the ABI.
//
// Cooley-Tukey with n = 18 n1 + n2, k = k1 + 16 k2:
//   X[k1 + 16 k2] = sum_{n2} W288^{n2 k1} ( sum_{n1} x[18 n1 + n2] W16^{n1 k1} ) W18^{n2 k2}
// step 1: 18 lanes, lane n2 holds x[18 n1 + n2] (n1 = 0..15), a 16-point transform in its registers, the twiddle
//         W288^{n2 k1}, stored to LDS position 18 k1 + n2 -- the lane's own 16 positions, so no hazard;
// step 2: 16 lanes, lane k1 reads its 18 contiguous values, an 18-point transform in registers, result k1 + 16 k2.
// Two LDS round trips per transform where the radix-3/3/4/4/2 passes of ldsfft.hpp take ten, ~220 instead of ~1000
// instructions per transform.  A 64-lane wave carries THREE transforms (54 lanes busy in step 1, 48 in step 2).
#pragma once
#ifdef __HIPCC__
#define F288_HD __device__ __forceinline__
#else
#define F288_HD inline
#endif

namespace f288 {
struct C2 {
    double x, y;
};
F288_HD C2 add(C2 a, C2 b) { return C2{a.x + b.x, a.y + b.y}; }
F288_HD C2 sub(C2 a, C2 b) { return C2{a.x - b.x, a.y - b.y}; }
// a * (c - i s) forward, a * (c + i s) inverse:  (c, s) = (cos, sin) of the positive angle
template <bool INV>
F288_HD C2 rot(C2 a, double c, double s) {
    return INV ? C2{__builtin_fma(a.x, c, -(a.y * s)), __builtin_fma(a.x, s, a.y * c)}
               : C2{__builtin_fma(a.x, c, a.y * s), __builtin_fma(a.y, c, -(a.x * s))};
}
// a * (-i) forward, a * (+i) inverse
template <bool INV>
F288_HD C2 rot90(C2 a) {
    return INV ? C2{-a.y, a.x} : C2{a.y, -a.x};
}

constexpr double SIN60 = 0.86602540378443864676;
constexpr double C8 = 0.92387953251128675613, S8 = 0.38268343236508977173, R2 = 0.70710678118654752440;  // pi/8, pi/4
constexpr double C20 = 0.93969262078590838405, S20 = 0.34202014332566873304;                               // 2 pi / 18
constexpr double C40 = 0.76604444311897803520, S40 = 0.64278760968653932632;
constexpr double C80 = 0.17364817766693034885, S80 = 0.98480775301220805937;

template <bool INV>
F288_HD void dft3(C2& a, C2& b, C2& c) {
    const C2 sm = add(b, c), df = sub(b, c);
    const C2 md = C2{__builtin_fma(-0.5, sm.x, a.x), __builtin_fma(-0.5, sm.y, a.y)};
    const double rx = (INV ? -SIN60 : SIN60) * df.y, ry = (INV ? SIN60 : -SIN60) * df.x;
    a = add(a, sm);
    b = C2{md.x + rx, md.y + ry};
    c = C2{md.x - rx, md.y - ry};
}
template <bool INV>
F288_HD void dft4(C2& a, C2& b, C2& c, C2& d) {
    const C2 t0 = add(a, c), t1 = sub(a, c), t2 = add(b, d), t3 = rot90<INV>(sub(b, d));
    a = add(t0, t2), b = add(t1, t3), c = sub(t0, t2), d = sub(t1, t3);
}

// v[0..8] -> the 9-point transform Y; Y[j1 + 3 j2] is left in v[3 j1 + j2]
template <bool INV>
F288_HD void dft9(C2* v) {
#pragma unroll
    for (int m2 = 0; m2 < 3; ++m2) dft3<INV>(v[m2], v[3 + m2], v[6 + m2]);
    v[4] = rot<INV>(v[4], C40, S40);   // W9^1
    v[5] = rot<INV>(v[5], C80, S80);   // W9^2
    v[7] = rot<INV>(v[7], C80, S80);   // W9^2
    v[8] = rot<INV>(v[8], -C20, S20);  // W9^4: 160 degrees
#pragma unroll
    for (int j1 = 0; j1 < 3; ++j1) dft3<INV>(v[3 * j1], v[3 * j1 + 1], v[3 * j1 + 2]);
}
// index of Y[k] in dft9's output
F288_HD int dft9_at(int k) { return 3 * (k % 3) + k / 3; }

// v[0..17] -> the 18-point transform X; X[k1 + 2 k2] is left in v[9 k1 + dft9_at(k2)]
template <bool INV>
F288_HD void dft18(C2* v) {
#pragma unroll
    for (int n2 = 0; n2 < 9; ++n2) {
        const C2 a = v[n2], b = v[9 + n2];
        v[n2] = add(a, b);
        v[9 + n2] = sub(a, b);
    }
    // W18^{n2}, n2 = 1..8: 20, 40, 60, 80, 100, 120, 140, 160 degrees
    v[10] = rot<INV>(v[10], C20, S20);
    v[11] = rot<INV>(v[11], C40, S40);
    v[12] = rot<INV>(v[12], 0.5, SIN60);
    v[13] = rot<INV>(v[13], C80, S80);
    v[14] = rot<INV>(v[14], -C80, S80);
    v[15] = rot<INV>(v[15], -0.5, SIN60);
    v[16] = rot<INV>(v[16], -C40, S40);
    v[17] = rot<INV>(v[17], -C20, S20);
    dft9<INV>(v);
    dft9<INV>(v + 9);
}
F288_HD int dft18_at(int k) { return 9 * (k & 1) + dft9_at(k >> 1); }

// v[0..15] -> the 16-point transform X; X[k1 + 4 k2] is left in v[4 k1 + k2]
template <bool INV>
F288_HD void dft16(C2* v) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // A[k1][n2] *= W16^{n2 k1}
    v[5] = rot<INV>(v[5], C8, S8);     // 1
    v[6] = rot<INV>(v[6], R2, R2);     // 2
    v[7] = rot<INV>(v[7], S8, C8);     // 3
    v[9] = rot<INV>(v[9], R2, R2);     // 2
    v[10] = rot90<INV>(v[10]);         // 4
    v[11] = rot<INV>(v[11], -R2, R2);  // 6
    v[13] = rot<INV>(v[13], S8, C8);   // 3
    v[14] = rot<INV>(v[14], -R2, R2);  // 6
    v[15] = rot<INV>(v[15], -C8, -S8); // 9: 202.5 degrees
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
F288_HD int dft16_at(int k) { return 4 * (k & 3) + (k >> 2); }

// ---- 320 = 16 x 20 and 384 = 16 x 24: the second factors
constexpr double C18 = 0.95105651629515357212, S18 = 0.30901699437494742410;  // also sin 72, cos 72
constexpr double C36 = 0.80901699437494742410, S36 = 0.58778525229247312917;  // also sin 54, cos 54
constexpr double C15 = 0.96592582628906828675, S15 = 0.25881904510252076235;

template <bool INV>
F288_HD void dft5(C2& x0, C2& x1, C2& x2, C2& x3, C2& x4) {
    const C2 t1 = add(x1, x4), t2 = add(x2, x3), t3 = sub(x1, x4), t4 = sub(x2, x3);
    // cos 72 = S18, cos 144 = -C36, sin 72 = C18, sin 144 = S36
    const C2 m1 = C2{__builtin_fma(S18, t1.x, __builtin_fma(-C36, t2.x, x0.x)), __builtin_fma(S18, t1.y, __builtin_fma(-C36, t2.y, x0.y))};
    const C2 m2 = C2{__builtin_fma(-C36, t1.x, __builtin_fma(S18, t2.x, x0.x)), __builtin_fma(-C36, t1.y, __builtin_fma(S18, t2.y, x0.y))};
    const C2 s1 = C2{__builtin_fma(C18, t3.x, S36 * t4.x), __builtin_fma(C18, t3.y, S36 * t4.y)};
    const C2 s2 = C2{__builtin_fma(S36, t3.x, -(C18 * t4.x)), __builtin_fma(S36, t3.y, -(C18 * t4.y))};
    // forward: X1 = m1 - i s1, X4 = m1 + i s1, X2 = m2 - i s2, X3 = m2 + i s2;  -i z = (z.y, -z.x)
    const C2 r1 = rot90<INV>(s1), r2 = rot90<INV>(s2);
    x0 = add(x0, add(t1, t2));
    x1 = add(m1, r1), x4 = sub(m1, r1);
    x2 = add(m2, r2), x3 = sub(m2, r2);
}
// v[0..5] -> Y; Y[j1 + 2 j2] is left in v[3 j1 + j2]
template <bool INV>
F288_HD void dft6(C2* v) {
#pragma unroll
    for (int m2 = 0; m2 < 3; ++m2) {
        const C2 a = v[m2], b = v[3 + m2];
        v[m2] = add(a, b), v[3 + m2] = sub(a, b);
    }
    v[4] = rot<INV>(v[4], 0.5, SIN60);   // W6^1
    v[5] = rot<INV>(v[5], -0.5, SIN60);  // W6^2
    dft3<INV>(v[0], v[1], v[2]);
    dft3<INV>(v[3], v[4], v[5]);
}
F288_HD int dft6_at(int k) { return 3 * (k & 1) + (k >> 1); }

// v[0..19] -> X (n = 5 n1 + n2, k = k1 + 4 k2); X[k1 + 4 k2] is left in v[5 k1 + k2]
template <bool INV>
F288_HD void dft20(C2* v) {
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) dft4<INV>(v[n2], v[5 + n2], v[10 + n2], v[15 + n2]);
    // A[k1][n2] *= W20^{n2 k1}: 18 degrees per unit
    v[6] = rot<INV>(v[6], C18, S18);     // 1
    v[7] = rot<INV>(v[7], C36, S36);     // 2
    v[8] = rot<INV>(v[8], S36, C36);     // 3: 54
    v[9] = rot<INV>(v[9], S18, C18);     // 4: 72
    v[11] = rot<INV>(v[11], C36, S36);   // 2
    v[12] = rot<INV>(v[12], S18, C18);   // 4
    v[13] = rot<INV>(v[13], -S18, C18);  // 6: 108
    v[14] = rot<INV>(v[14], -C36, S36);  // 8: 144
    v[16] = rot<INV>(v[16], S36, C36);   // 3
    v[17] = rot<INV>(v[17], -S18, C18);  // 6
    v[18] = rot<INV>(v[18], -C18, S18);  // 9: 162
    v[19] = rot<INV>(v[19], -C36, -S36); // 12: 216
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft5<INV>(v[5 * k1], v[5 * k1 + 1], v[5 * k1 + 2], v[5 * k1 + 3], v[5 * k1 + 4]);
}
F288_HD int dft20_at(int k) { return 5 * (k & 3) + (k >> 2); }

// v[0..23] -> X (n = 6 n1 + n2, k = k1 + 4 k2); X[k1 + 4 k2] is left in v[6 k1 + dft6_at(k2)]
template <bool INV>
F288_HD void dft24(C2* v) {
#pragma unroll
    for (int n2 = 0; n2 < 6; ++n2) dft4<INV>(v[n2], v[6 + n2], v[12 + n2], v[18 + n2]);
    // A[k1][n2] *= W24^{n2 k1}: 15 degrees per unit
    v[7] = rot<INV>(v[7], C15, S15);        // 1
    v[8] = rot<INV>(v[8], SIN60, 0.5);      // 2: 30
    v[9] = rot<INV>(v[9], R2, R2);          // 3: 45
    v[10] = rot<INV>(v[10], 0.5, SIN60);    // 4: 60
    v[11] = rot<INV>(v[11], S15, C15);      // 5: 75
    v[13] = rot<INV>(v[13], SIN60, 0.5);    // 2
    v[14] = rot<INV>(v[14], 0.5, SIN60);    // 4
    v[15] = rot90<INV>(v[15]);              // 6: 90
    v[16] = rot<INV>(v[16], -0.5, SIN60);   // 8: 120
    v[17] = rot<INV>(v[17], -SIN60, 0.5);   // 10: 150
    v[19] = rot<INV>(v[19], R2, R2);        // 3
    v[20] = rot90<INV>(v[20]);              // 6
    v[21] = rot<INV>(v[21], -R2, R2);       // 9: 135
    v[22] = C2{-v[22].x, -v[22].y};         // 12: 180
    v[23] = rot<INV>(v[23], -R2, -R2);      // 15: 225
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft6<INV>(v + 6 * k1);
}
F288_HD int dft24_at(int k) { return 6 * (k & 3) + dft6_at(k >> 2); }

// the second factor M of a 16 x M frame: its transform and where that leaves output k
template <int M>
struct Second;
template <>
struct Second<18> {
    template <bool INV>
    static F288_HD void run(C2* v) { dft18<INV>(v); }
    static F288_HD int at(int k) { return dft18_at(k); }
};
template <>
struct Second<20> {
    template <bool INV>
    static F288_HD void run(C2* v) { dft20<INV>(v); }
    static F288_HD int at(int k) { return dft20_at(k); }
};
template <>
struct Second<24> {
    template <bool INV>
    static F288_HD void run(C2* v) { dft24<INV>(v); }
    static F288_HD int at(int k) { return dft24_at(k); }
};

#ifdef __HIPCC__
// ---- a transform by ONE 32-lane group in LDS (the lane mapping of ldsfft.hpp's passes: two groups per wave), for the
// kernels whose other stages use all 32 lanes.  buf: the sequence, transformed in place; tw: e^{-2 pi i k / 288}, k < 288;
// t: lane within the group.  The caller has made buf visible to the group (group_sync) and gets it back visible.
__device__ __forceinline__ void group_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int M, bool INV>
__device__ __forceinline__ void fft16_group(C2* __restrict__ buf, const C2* __restrict__ tw, int t) {
    C2 v[M];
    if (t < M) {  // 16-point transforms over n1 of x[M n1 + t], twiddle, back to the lane's own positions
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = buf[M * n1 + t];
        dft16<INV>(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            C2 a = v[dft16_at(q)];
            if (q > 0) {
                const C2 e = tw[t * q];
                a = rot<INV>(a, e.x, -e.y);
            }
            buf[M * q + t] = a;
        }
    }
    group_fence();
    if (t < 16) {
#pragma unroll
        for (int q = 0; q < M; ++q) v[q] = buf[M * t + q];
        Second<M>::template run<INV>(v);
    }
    group_fence();
    if (t < 16) {
#pragma unroll
        for (int q = 0; q < M; ++q) buf[t + 16 * q] = v[Second<M>::at(q)];
    }
    group_fence();
}
// 144 = 16 x 9 (the half-length sequence of a real row of 288): n = 9 n1 + n2, k = k1 + 16 k2; W144^m = tw[2 m]
template <bool INV>
__device__ __forceinline__ void fft144_group(C2* __restrict__ buf, const C2* __restrict__ tw, int t) {
    C2 v[16];
    if (t < 9) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = buf[9 * n1 + t];
        dft16<INV>(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            C2 a = v[dft16_at(q)];
            if (q > 0) {
                const C2 e = tw[2 * t * q];
                a = rot<INV>(a, e.x, -e.y);
            }
            buf[9 * q + t] = a;
        }
    }
    group_fence();
    if (t < 16) {
#pragma unroll
        for (int q = 0; q < 9; ++q) v[q] = buf[9 * t + q];
        dft9<INV>(v);
    }
    group_fence();
    if (t < 16) {
#pragma unroll
        for (int q = 0; q < 9; ++q) buf[t + 16 * q] = v[dft9_at(q)];
    }
    group_fence();
}
#endif
}  // namespace f288
