// Test harness (NOT part of the product): exposes the scalar solvers of getdist_amd/csrc/solvers.hpp to ctypes with a
// Python callback as the function, so that tests can compare them evaluation by evaluation with scipy's fsolve /
// brentq / TNC driving the very same Python function.
#include "../../getdist_amd/csrc/solvers.hpp"

extern "C" {

typedef double (*gdt_fcn)(double x, int* fail);

int gdt_hybrd1(gdt_fcn f, double x0, double xtol, int maxfev, double factor, double* x_out, int* nfev_out) {
    auto fcn = [&](double x, bool* fail) {
        int fl = 0;
        const double v = f(x, &fl);
        if (fl) *fail = true;
        return v;
    };
    const gdsolve::HybrdResult r = gdsolve::hybrd1(fcn, x0, xtol, maxfev, factor);
    *x_out = r.x;
    *nfev_out = r.nfev;
    return r.info;
}

int gdt_brentq(gdt_fcn f, double xa, double xb, double xtol, double rtol, int maxiter, double* x_out, int* nfev_out) {
    auto fcn = [&](double x, bool* fail) {
        int fl = 0;
        const double v = f(x, &fl);
        if (fl) *fail = true;
        return v;
    };
    const gdsolve::BrentResult r = gdsolve::brentq(fcn, xa, xb, xtol, rtol, maxiter);
    *x_out = r.x;
    *nfev_out = r.nfev;
    return r.status;
}

typedef double (*gdt_fcn_nd)(const double* x, int n, int* fail);

// scipy.optimize.minimize(fun, x0, method="TNC", bounds=zip(low, up)) for n <= 3; returns the TNC return code
// (7 = aborted: an exception in fun or x0 outside the bounds), *success as scipy's OptimizeResult.success
int gdt_tnc(gdt_fcn_nd f, int n, const double* x0, const double* low, const double* up, double* x_out, int* success,
            int* nfev_total, int* niter) {
    auto fcn = [&](const double* x, bool* fail) {
        int fl = 0;
        const double v = f(x, n, &fl);
        if (fl) *fail = true;
        return v;
    };
    gdsolve::Tnc<decltype(fcn)> tnc(fcn, n);
    const gdsolve::TncResult r = tnc.run(x0, low, up);
    for (int i = 0; i < n; ++i) x_out[i] = r.x[i];
    *success = r.success ? 1 : 0;
    *nfev_total = tnc.nfev_total;
    *niter = r.niter;
    return r.rc;
}

// KernelOptimizer2D.get_h in plain C++ arithmetic (no callback): psi = (p02, p20, p11, p00, p13, p31)
int gdt_get_h(const double* psi, double N, double corr_in, int do_corr, double* out3, int* nfev) {
    const gdsolve::GetHResult r = gdsolve::get_h(psi, N, corr_in, do_corr != 0);
    out3[0] = r.hx, out3[1] = r.hy, out3[2] = r.corr;
    *nfev = r.nfev;
    return r.status;
}

double gdt_amise(const double* cov, const double* p5, double N, double corr, int fixed) {
    gdsolve::Amise am;
    am.p40 = p5[0], am.p04 = p5[1], am.p22 = p5[2], am.p13 = p5[3], am.p31 = p5[4], am.N = N, am.corr = corr;
    am.fixed_corr = fixed != 0;
    bool fail = false;
    const double v = am(cov, &fail);
    return fail ? -1.0 : v;
}

}  // extern "C"
