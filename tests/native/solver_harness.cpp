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

}  // extern "C"
