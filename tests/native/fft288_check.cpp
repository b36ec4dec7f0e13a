// Host check of the register codelets of getdist_amd/csrc/fft288.hpp against the definition of the transform.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include "../../getdist_amd/csrc/fft288.hpp"
using namespace f288;
typedef std::complex<long double> cl;
template <int N, bool INV, void (*F)(C2*), int (*AT)(int)>
static double check(unsigned seed) {
    C2 v[N];
    cl x[N];
    srand(seed);
    for (int i = 0; i < N; ++i) {
        v[i] = C2{rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
        x[i] = cl(v[i].x, v[i].y);
    }
    F(v);
    const long double pi = 3.141592653589793238462643383279502884L;
    double worst = 0;
    for (int k = 0; k < N; ++k) {
        cl acc = 0;
        for (int n = 0; n < N; ++n) {
            const long double a = (INV ? 2 : -2) * pi * (long double)((n * k) % N) / N;
            acc += x[n] * cl(cosl(a), sinl(a));
        }
        const C2 got = v[AT(k)];
        worst = fmax(worst, fmax(fabs((double)(acc.real() - got.x)), fabs((double)(acc.imag() - got.y))));
    }
    return worst;
}

template <int M>
static double compose() {
    const int N = 16 * M;
    static C2 x[16 * 24], a[16 * 24], out[16 * 24];
    srand(7 + M);
    for (int i = 0; i < N; ++i) x[i] = C2{rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    const long double pi = 3.141592653589793238462643383279502884L;
    double worst = 0;
    for (int inv = 0; inv < 2; ++inv) {
        for (int n2 = 0; n2 < M; ++n2) {
            C2 v[16];
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = x[M * n1 + n2];
            if (inv) dft16<true>(v); else dft16<false>(v);
            for (int k1 = 0; k1 < 16; ++k1) {
                const long double ang = 2 * pi * (long double)(n2 * k1) / N;
                const double c = (double)cosl(ang), s = (double)sinl(ang);
                a[M * k1 + n2] = inv ? rot<true>(v[dft16_at(k1)], c, s) : rot<false>(v[dft16_at(k1)], c, s);
            }
        }
        for (int k1 = 0; k1 < 16; ++k1) {
            C2 v[M];
            for (int n2 = 0; n2 < M; ++n2) v[n2] = a[M * k1 + n2];
            if (inv) Second<M>::template run<true>(v); else Second<M>::template run<false>(v);
            for (int k2 = 0; k2 < M; ++k2) out[k1 + 16 * k2] = v[Second<M>::at(k2)];
        }
        for (int k = 0; k < N; ++k) {
            cl acc = 0;
            for (int n = 0; n < N; ++n) {
                const long double ang = (inv ? 2 : -2) * pi * (long double)((n * k) % N) / N;
                acc += cl(x[n].x, x[n].y) * cl(cosl(ang), sinl(ang));
            }
            worst = fmax(worst, fmax(fabs((double)(acc.real() - out[k].x)), fabs((double)(acc.imag() - out[k].y))));
        }
    }
    return worst;
}
int main() {
    double w = 0;
    for (unsigned s = 1; s <= 50; ++s) {
        w = fmax(w, check<9, false, dft9<false>, dft9_at>(s));
        w = fmax(w, check<9, true, dft9<true>, dft9_at>(s));
        w = fmax(w, check<16, false, dft16<false>, dft16_at>(s));
        w = fmax(w, check<16, true, dft16<true>, dft16_at>(s));
        w = fmax(w, check<18, false, dft18<false>, dft18_at>(s));
        w = fmax(w, check<18, true, dft18<true>, dft18_at>(s));
        w = fmax(w, check<6, false, dft6<false>, dft6_at>(s));
        w = fmax(w, check<6, true, dft6<true>, dft6_at>(s));
        w = fmax(w, check<20, false, dft20<false>, dft20_at>(s));
        w = fmax(w, check<20, true, dft20<true>, dft20_at>(s));
        w = fmax(w, check<24, false, dft24<false>, dft24_at>(s));
        w = fmax(w, check<24, true, dft24<true>, dft24_at>(s));
    }
    // the whole 16 x M transforms as the kernels compose them
    double w288 = 0;
    w288 = fmax(w288, compose<18>());
    w288 = fmax(w288, compose<20>());
    w288 = fmax(w288, compose<24>());
    printf("codelets worst abs error %.3e; 16 x {18, 20, 24} compositions worst abs error %.3e\n", w, w288);
    return (w < 1e-14 && w288 < 1e-13) ? 0 : 1;
}
