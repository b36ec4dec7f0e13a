// Scalar root finders of the bandwidth selection, written so that the SAME source runs inside a HIP kernel
// (every thread of a block executes the control flow redundantly; the function object is a block-collective
// evaluation) and in a plain C++ harness (tests/native) where it is checked evaluation by evaluation against
// scipy.optimize.fsolve / brentq.
//
// The reference's results are wherever these iterations stop (xtol = hfrac/20 for the 1D bandwidth), so the
// iteration path -- not just the root -- defines parity.
#pragma once
#include <math.h>

#ifndef GD_HD
#ifdef __HIPCC__
#define GD_HD __host__ __device__
#else
#define GD_HD
#endif
#endif

namespace gdsolve {

constexpr double EPSMCH = 2.220446049250313e-16;  // dpmpar(1)

// ---------------------------------------------------------------------------------------------------------------
// MINPACK hybrd specialised to ONE unknown, as scipy.optimize.fsolve drives it: mode = 1 (internal scaling),
// nprint = 0, ml = mu = 0, epsfcn = machine epsilon (kde_bandwidth.py:123 passes xtol and factor only).
// With n = 1 the QR factorisation of the forward-difference Jacobian a is  Q = -1, R = -a  (qrfac's Householder
// reflection of a 1-vector), (Q^T f) = -f, the dogleg step is either the Newton step or the trust-region boundary
// along it, Broyden's rank-one update adds (v u) to R (r1updt) and the rotations of r1mpyq are empty.  Every
// arithmetic expression below keeps hybrd.f / dogleg.f / fdjac1.f's operation order so that rounding agrees.
//
// fcn(x, &fail) returns f(x); setting fail aborts (scipy: an exception raised inside the callback).
// Returns info (1 = converged ... 5, hybrd.f; -1 = aborted by fcn); *x_out receives the final iterate.
struct HybrdResult {
    double x;
    int info, nfev;
};

template <class Fcn>
GD_HD HybrdResult hybrd1(Fcn&& fcn, double x, double xtol, int maxfev, double factor) {
    const double p1 = 0.1, p5 = 0.5, p001 = 1.0e-3, p0001 = 1.0e-4;
    HybrdResult res;
    res.info = 0;
    res.nfev = 0;
    bool fail = false;
    double fvec = fcn(x, &fail);
    res.nfev = 1;
    res.x = x;
    if (fail) {
        res.info = -1;
        return res;
    }
    double fnorm = fabs(fvec);
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double diag = 0, delta = 0, xnorm = 0;
    int info = 0;
    for (;;) {  // outer loop: (re)compute the Jacobian by forward differences
        bool jeval = true;
        // fdjac1
        const double eps = sqrt(fmax(EPSMCH, EPSMCH));
        double h = eps * fabs(x);
        if (h == 0.0) h = eps;
        const double fh = fcn(x + h, &fail);
        res.nfev += 1;
        if (fail) {
            info = -1;
            break;
        }
        const double a = (fh - fvec) / h;
        // qrfac (1 x 1): column norm |a|; R = -a (rdiag), Q = -1
        const double acnorm = fabs(a);
        double r = -a;
        if (acnorm == 0.0) r = 0.0;  // ajnorm == 0: rdiag = -ajnorm = -0
        if (iter == 1) {
            diag = acnorm;
            if (acnorm == 0.0) diag = 1.0;
            xnorm = fabs(diag * x);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        // (Q^T f): Householder vector 2 on a 1-vector:  qtf = f + 2 * (-(2 f) / 2)  [only if the Jacobian is nonzero]
        double qtf = fvec;
        if (acnorm != 0.0) {
            const double sum = 2.0 * qtf;
            const double temp = -sum / 2.0;
            qtf = qtf + 2.0 * temp;
        }
        // qform leaves Q = -1 for a nonzero column, +1 for a zero column
        const double q = (acnorm != 0.0) ? -1.0 : 1.0;
        diag = fmax(diag, acnorm);
        bool leave = false;
        for (;;) {  // inner loop
            // ---- dogleg (n = 1)
            double gn;  // Gauss-Newton direction
            {
                double temp = r;
                if (temp == 0.0) {
                    temp = EPSMCH * fabs(r);
                    if (temp == 0.0) temp = EPSMCH;
                }
                gn = qtf / temp;
            }
            double p = gn;
            const double qnorm = fabs(diag * gn);
            if (!(qnorm <= delta)) {
                double wa1 = (r * qtf) / diag;
                const double gnorm = fabs(wa1);
                double sgnorm = 0.0;
                double alpha = delta / qnorm;
                if (gnorm != 0.0) {
                    wa1 = (wa1 / gnorm) / diag;
                    const double temp = fabs(r * wa1);
                    sgnorm = (gnorm / temp) / temp;
                    alpha = 0.0;
                    if (sgnorm < delta) {
                        const double bnorm = fabs(qtf);
                        double t = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
                        const double dq = delta / qnorm, sd = sgnorm / delta;
                        t = t - dq * (sd * sd) + sqrt((t - dq) * (t - dq) + (1.0 - dq * dq) * (1.0 - sd * sd));
                        alpha = (dq * (1.0 - sd * sd)) / t;
                    }
                }
                const double temp = (1.0 - alpha) * fmin(sgnorm, delta);
                p = temp * wa1 + alpha * gn;
            }
            // ---- trial point
            const double wa1 = -p;
            const double xt = x + wa1;
            const double pnorm = fabs(diag * wa1);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double f1 = fcn(xt, &fail);
            res.nfev += 1;
            if (fail) {
                info = -1;
                leave = true;
                break;
            }
            const double fnorm1 = fabs(f1);
            double actred = -1.0;
            if (fnorm1 < fnorm) {
                const double t = fnorm1 / fnorm;
                actred = 1.0 - t * t;
            }
            const double wa3 = qtf + r * wa1;
            const double tn = fabs(wa3);
            double prered = 0.0;
            if (tn < fnorm) {
                const double t = tn / fnorm;
                prered = 1.0 - t * t;
            }
            double ratio = 0.0;
            if (prered > 0.0) ratio = actred / prered;
            if (ratio >= p1) {
                ncfail = 0;
                ncsuc += 1;
                if (ratio >= p5 || ncsuc > 1) delta = fmax(delta, pnorm / p5);
                if (fabs(ratio - 1.0) <= p1) delta = pnorm / p5;
            } else {
                ncsuc = 0;
                ncfail += 1;
                delta = p5 * delta;
            }
            if (ratio >= p0001) {  // successful iteration
                x = xt;
                xnorm = fabs(diag * x);
                fvec = f1;
                fnorm = fnorm1;
                iter += 1;
            }
            nslow1 += 1;
            if (actred >= p001) nslow1 = 0;
            if (jeval) nslow2 += 1;
            if (actred >= p1) nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0.0) info = 1;
            if (info != 0) {
                leave = true;
                break;
            }
            if (res.nfev >= maxfev) info = 2;
            if (p1 * fmax(p1 * delta, pnorm) <= EPSMCH * xnorm) info = 3;
            if (nslow2 == 5) info = 4;
            if (nslow1 == 10) info = 5;
            if (info != 0) {
                leave = true;
                break;
            }
            if (ncfail == 2) break;  // recompute the Jacobian
            // ---- Broyden rank-one update: sum = Q^T f1; v = (sum - wa3)/pnorm; u = diag*((diag*wa1)/pnorm)
            const double sum = q * f1;
            const double v = (sum - wa3) / pnorm;
            const double u = diag * ((diag * wa1) / pnorm);
            if (ratio >= p0001) qtf = sum;
            r = r + v * u;  // r1updt, n = 1
            jeval = false;
        }
        if (leave) break;
    }
    res.x = x;
    res.info = info;
    return res;
}

// ---------------------------------------------------------------------------------------------------------------
// scipy.optimize.brentq (Zeros/brentq.c): classic Brent with inverse quadratic extrapolation.
// status: 0 converged, -1 f(a) and f(b) have the same sign, -2 no convergence in maxiter, -3 aborted by fcn / NaN.
struct BrentResult {
    double x;
    int status, nfev;
};

template <class Fcn>
GD_HD BrentResult brentq(Fcn&& fcn, double xa, double xb, double xtol, double rtol, int maxiter) {
    BrentResult res;
    res.nfev = 0;
    bool fail = false;
    double xpre = xa, xcur = xb, xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
    double fpre = fcn(xpre, &fail);
    double fcur = fail ? 0.0 : fcn(xcur, &fail);
    res.nfev = 2;
    res.x = 0.0;
    if (fail || fpre != fpre || fcur != fcur) {
        res.status = -3;
        return res;
    }
    if (fpre == 0) {
        res.x = xpre, res.status = 0;
        return res;
    }
    if (fcur == 0) {
        res.x = xcur, res.status = 0;
        return res;
    }
    if (signbit(fpre) == signbit(fcur)) {
        res.status = -1;
        return res;
    }
    res.status = -2;
    for (int it = 0; it < maxiter; ++it) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur, xcur = xblk, xblk = xpre;
            fpre = fcur, fcur = fblk, fblk = fpre;
        }
        const double delta = (xtol + rtol * fabs(xcur)) / 2;
        const double sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) {
            res.x = xcur;
            res.status = 0;
            return res;
        }
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                const double dpre = (fpre - fcur) / (xpre - xcur);
                const double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) {
                spre = scur;
                scur = stry;
            } else {
                spre = sbis;
                scur = sbis;
            }
        } else {
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta)
            xcur += scur;
        else
            xcur += (sbis > 0 ? delta : -delta);
        fcur = fcn(xcur, &fail);
        res.nfev += 1;
        if (fail || fcur != fcur) {
            res.status = -3;
            return res;
        }
    }
    return res;
}

}  // namespace gdsolve
