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
int main() {
    double w = 0;
    for (unsigned s = 1; s <= 50; ++s) {
        w = fmax(w, check<9, false, dft9<false>, dft9_at>(s));
        w = fmax(w, check<9, true, dft9<true>, dft9_at>(s));
        w = fmax(w, check<16, false, dft16<false>, dft16_at>(s));
        w = fmax(w, check<16, true, dft16<true>, dft16_at>(s));
        w = fmax(w, check<18, false, dft18<false>, dft18_at>(s));
        w = fmax(w, check<18, true, dft18<true>, dft18_at>(s));
    }
    // the whole 288-point transform as the kernels compose it
    const int N = 288;
    static C2 x[N], a[N], out[N];
    srand(7);
    for (int i = 0; i < N; ++i) x[i] = C2{rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    const long double pi = 3.141592653589793238462643383279502884L;
    double w288 = 0;
    for (int inv = 0; inv < 2; ++inv) {
        for (int n2 = 0; n2 < 18; ++n2) {
            C2 v[16];
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = x[18 * n1 + n2];
            if (inv) dft16<true>(v); else dft16<false>(v);
            for (int k1 = 0; k1 < 16; ++k1) {
                const long double ang = 2 * pi * (long double)(n2 * k1) / N;
                const double c = (double)cosl(ang), s = (double)sinl(ang);
                a[18 * k1 + n2] = inv ? rot<true>(v[dft16_at(k1)], c, s) : rot<false>(v[dft16_at(k1)], c, s);
            }
        }
        for (int k1 = 0; k1 < 16; ++k1) {
            C2 v[18];
            for (int n2 = 0; n2 < 18; ++n2) v[n2] = a[18 * k1 + n2];
            if (inv) dft18<true>(v); else dft18<false>(v);
            for (int k2 = 0; k2 < 18; ++k2) out[k1 + 16 * k2] = v[dft18_at(k2)];
        }
        for (int k = 0; k < N; ++k) {
            cl acc = 0;
            for (int n = 0; n < N; ++n) {
                const long double ang = (inv ? 2 : -2) * pi * (long double)((n * k) % N) / N;
                acc += cl(x[n].x, x[n].y) * cl(cosl(ang), sinl(ang));
            }
            w288 = fmax(w288, fmax(fabs((double)(acc.real() - out[k].x)), fabs((double)(acc.imag() - out[k].y))));
        }
    }
    printf("codelets worst abs error %.3e; 288-point composition worst abs error %.3e\n", w, w288);
    return (w < 1e-14 && w288 < 1e-13) ? 0 : 1;
}
