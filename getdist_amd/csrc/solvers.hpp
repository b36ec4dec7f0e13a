// Scalar root finders of the bandwidth selection, written so that the SAME source runs inside a HIP kernel
// (every thread of a block executes the control flow redundantly; the function object is a block-collective
// evaluation) and in a plain C++ harness (tests/native) where it is checked evaluation by evaluation against
// scipy.optimize.fsolve / brentq.
//
// The reference's results are wherever these iterations stop (xtol = hfrac/20 for the 1D bandwidth), so the
// iteration path -- not just the root -- defines parity.
#pragma once
#include <math.h>
#ifdef GD_TNC_DEBUG
#include <stdio.h>
#define GD_DBG(...) printf(__VA_ARGS__)
#else
#define GD_DBG(...)
#endif

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


// ---------------------------------------------------------------------------------------------------------------
// scipy.optimize.minimize(method="TNC", bounds=...) with the default finite-difference gradient, for n <= 3 unknowns:
//   * scipy's ScalarFunction: f(x) then one forward difference per unknown with absolute step 1e-8, the step
//     flipped where x + h would leave the bounds (_numdiff.py _adjust_scheme_to_bounds), df / dx with dx recomputed
//     as (x + h) - x; f and gradient memoised for a repeated x;
//   * tnc.c 1.3 (J.S. Roy's C version of S.G. Nash's truncated Newton code) with the defaults scipy passes: variable
//     scaling by the bound widths, function rescaling, preconditioned CG on finite-difference Hessian-vector products,
//     the Gill-Murray line search getptc, active-set handling of the bounds.
// The reference accepts or rejects the optimiser's result by comparing AMISE values (kde_bandwidth.py:276-302), and
// the result is wherever this iteration stops -- so, as for hybrd above, the control flow and the operation order
// are kept, and tests/test_native_solvers.py compares the sequence of evaluation points with scipy's, bit for bit.
constexpr int TNC_MAXN = 3;
constexpr double TNC_HUGE = HUGE_VAL;

enum TncRc {
    TNC_MINRC = -3, TNC_ENOMEM = -3, TNC_EINVAL = -2, TNC_INFEASIBLE = -1, TNC_LOCALMINIMUM = 0, TNC_FCONVERGED = 1,
    TNC_XCONVERGED = 2, TNC_MAXFUN = 3, TNC_LSFAIL = 4, TNC_CONSTANT = 5, TNC_NOPROGRESS = 6, TNC_USERABORT = 7
};

struct TncResult {
    double x[TNC_MAXN];
    int rc, nfev, niter;
    bool success;  // scipy: -1 < rc < 3
};

template <class Fcn>
struct Tnc {
    Fcn& fcn;
    int n;
    double low[TNC_MAXN], up[TNC_MAXN];
    // ScalarFunction state
    double sx[TNC_MAXN], sf, sg[TNC_MAXN];
    bool have;
    int nfev_total;

    GD_HD Tnc(Fcn& f, int n_) : fcn(f), n(n_), sf(0), have(false), nfev_total(0) {}

    // --- scipy ScalarFunction.fun_and_grad with '2-point' differences, abs_step = 1e-8, bounds
    // Device form (round 5): the n + 1 function values of one call -- f(x) and one forward difference per unknown -- are
    // independent, and every evaluation of TNC (iterates, line-search trials, the finite-difference Hessian-vector
    // products) goes through this function.  The FOUR lowest lanes of the wavefront run the whole optimiser in lock step
    // on identical state; here lane q evaluates its own point (lane 0: x, lane i + 1: x + h_i e_i) and the values are
    // exchanged by wave shuffles: one AMISE evaluation (two pow, a square root, a division) per call instead of n + 1 in
    // sequence.  Every value is computed by the same operations on the same operands as in the sequential form, the
    // fail / bound tests are applied in its order: the same bits, the same evaluation count (tests/test_gpu_primitives.py
    // compares with the host build evaluation by evaluation as before).  The caller keeps lanes 0..3 active and converged.
    GD_HD int function(const double* x, double* f, double* g) {
        bool same = have;
        for (int i = 0; i < n && same; ++i) same = (x[i] == sx[i]);
#ifdef __HIP_DEVICE_COMPILE__
        if (!same) {
            for (int i = 0; i < n; ++i) sx[i] = x[i];
            have = false;
            double h[TNC_MAXN];
            for (int i = 0; i < n; ++i) {
                double hi = 1e-8;
                const double dx = (sx[i] + hi) - sx[i];
                if (dx == 0.0) hi = EPSMCH_SQRT() * ((sx[i] >= 0) ? 1.0 : -1.0) * fmax(1.0, fabs(sx[i]));
                const double lower_dist = sx[i] - low[i], upper_dist = up[i] - sx[i];
                const double xt = sx[i] + hi;
                const bool violated = (xt < low[i]) || (xt > up[i]);
                const bool fitting = fabs(hi) <= fmax(lower_dist, upper_dist);
                double ha = hi;
                if (violated && fitting) ha = -hi;
                if (!fitting) ha = (upper_dist >= lower_dist) ? upper_dist : -lower_dist;
                h[i] = ha;
            }
            const int me = (int)(threadIdx.x & 3) - 1;  // -1: the point itself
            double xp[TNC_MAXN];
            for (int q = 0; q < n; ++q) xp[q] = sx[q];
            for (int q = 0; q < n; ++q)
                if (q == me) xp[q] = sx[q] + h[q];
            bool my_fail = false;
            const double my_f = fcn(xp, &my_fail);
            const int my_fail_i = my_fail ? 1 : 0;
            sf = __shfl(my_f, 0, 4);
            nfev_total += 1;
            if (__shfl(my_fail_i, 0, 4)) return 1;
            for (int i = 0; i < n; ++i) {
                if (sx[i] < low[i] || sx[i] > up[i]) return 1;  // "`x0` violates bound constraints."
            }
            for (int i = 0; i < n; ++i) {
                const double f1 = __shfl(my_f, i + 1, 4);
                const int fl = __shfl(my_fail_i, i + 1, 4);
                nfev_total += 1;
                if (fl) return 1;
                const double dx = (sx[i] + h[i]) - sx[i];
                sg[i] = (f1 - sf) / dx;
            }
            have = true;
        }
#else
        if (!same) {
            bool fail = false;
            for (int i = 0; i < n; ++i) sx[i] = x[i];
            have = false;
            sf = fcn(sx, &fail);
            nfev_total += 1;
            if (fail) return 1;
            for (int i = 0; i < n; ++i) {
                if (sx[i] < low[i] || sx[i] > up[i]) return 1;  // "`x0` violates bound constraints."
            }
            double h[TNC_MAXN];
            for (int i = 0; i < n; ++i) {
                double hi = 1e-8;
                const double dx = (sx[i] + hi) - sx[i];
                if (dx == 0.0) hi = EPSMCH_SQRT() * ((sx[i] >= 0) ? 1.0 : -1.0) * fmax(1.0, fabs(sx[i]));
                // _adjust_scheme_to_bounds, 1-sided, one step
                const double lower_dist = sx[i] - low[i], upper_dist = up[i] - sx[i];
                const double xt = sx[i] + hi;
                const bool violated = (xt < low[i]) || (xt > up[i]);
                const bool fitting = fabs(hi) <= fmax(lower_dist, upper_dist);
                double ha = hi;
                if (violated && fitting) ha = -hi;
                if (!fitting) ha = (upper_dist >= lower_dist) ? upper_dist : -lower_dist;
                h[i] = ha;
            }
            for (int i = 0; i < n; ++i) {
                double xp[TNC_MAXN];
                for (int q = 0; q < n; ++q) xp[q] = sx[q];
                xp[i] = sx[i] + h[i];
                const double dx = xp[i] - sx[i];
                const double f1 = fcn(xp, &fail);
                nfev_total += 1;
                if (fail) return 1;
                sg[i] = (f1 - sf) / dx;
            }
            have = true;
        }
#endif
        *f = sf;
        for (int i = 0; i < n; ++i) g[i] = sg[i];
        return 0;
    }
    GD_HD static double EPSMCH_SQRT() { return 1.4901161193847656e-08; }

    // --- small vector helpers of tnc.c
    GD_HD double ddot1(const double* a, const double* b) const {
        double d = 0.0;
        for (int i = 0; i < n; ++i) d += a[i] * b[i];
        return d;
    }
    GD_HD double dnrm21(const double* dx) const {  // Euclidean norm with running rescaling against overflow (tnc.c dnrm21)
        double dssq = 1.0, dscale = 0.0;
        for (int i = 0; i < n; ++i) {
            if (dx[i] != 0.0) {
                const double dabsxi = fabs(dx[i]);
                if (dscale < dabsxi) {
                    const double ratio = dscale / dabsxi;
                    dssq = 1.0 + dssq * ratio * ratio;
                    dscale = dabsxi;
                } else {
                    const double ratio = dabsxi / dscale;
                    dssq += ratio * ratio;
                }
            }
        }
        return dscale * sqrt(dssq);
    }
    GD_HD void project(double* v, const int* pivot) const {
        for (int i = 0; i < n; ++i)
            if (pivot[i] != 0) v[i] = 0.0;
    }
    GD_HD void coercex(double* x) const {
        for (int i = 0; i < n; ++i) {
            if (x[i] < low[i])
                x[i] = low[i];
            else if (x[i] > up[i])
                x[i] = up[i];
        }
    }
    GD_HD void unscalex(double* x, const double* xscale, const double* xoffset) const {
        for (int i = 0; i < n; ++i) x[i] = x[i] * xscale[i] + xoffset[i];
    }
    GD_HD void scalex(double* x, const double* xscale, const double* xoffset) const {
        for (int i = 0; i < n; ++i)
            if (xscale[i] > 0.0) x[i] = (x[i] - xoffset[i]) / xscale[i];
    }
    GD_HD void scaleg(double* g, const double* xscale, double fscale) const {
        for (int i = 0; i < n; ++i) g[i] *= xscale[i] * fscale;
    }
    GD_HD void setConstraints(const double* x, int* pivot, const double* xscale, const double* xoffset) const {
        for (int i = 0; i < n; ++i) {
            if (xscale[i] == 0.0) {
                pivot[i] = 2;
            } else if (low[i] != -TNC_HUGE &&
                       (x[i] * xscale[i] + xoffset[i] - low[i] <= EPSMCH * 10.0 * (fabs(low[i]) + 1.0))) {
                pivot[i] = -1;
            } else if (up[i] != TNC_HUGE &&
                       (x[i] * xscale[i] + xoffset[i] - up[i] >= EPSMCH * 10.0 * (fabs(up[i]) + 1.0))) {
                pivot[i] = 1;
            } else {
                pivot[i] = 0;
            }
        }
    }
    GD_HD double stepMax(double step, const double* x, const double* dir, const int* pivot, const double* xscale,
                         const double* xoffset) const {
        for (int i = 0; i < n; ++i) {
            if (pivot[i] == 0 && dir[i] != 0.0) {
                if (dir[i] < 0.0) {
                    const double t = (low[i] - xoffset[i]) / xscale[i] - x[i];
                    if (t > step * dir[i]) step = t / dir[i];
                } else {
                    const double t = (up[i] - xoffset[i]) / xscale[i] - x[i];
                    if (t < step * dir[i]) step = t / dir[i];
                }
            }
        }
        return step;
    }
    GD_HD bool addConstraint(double* x, const double* p, int* pivot, const double* xscale, const double* xoffset) const {
        bool newcon = false;
        for (int i = 0; i < n; ++i) {
            if (pivot[i] == 0 && p[i] != 0.0) {
                if (p[i] < 0.0 && low[i] != -TNC_HUGE) {
                    const double tol = EPSMCH * 10.0 * (fabs(low[i]) + 1.0);
                    if (x[i] * xscale[i] + xoffset[i] - low[i] <= tol) {
                        pivot[i] = -1;
                        x[i] = (low[i] - xoffset[i]) / xscale[i];
                        newcon = true;
                    }
                } else if (up[i] != TNC_HUGE) {
                    const double tol = EPSMCH * 10.0 * (fabs(up[i]) + 1.0);
                    if (up[i] - (x[i] * xscale[i] + xoffset[i]) <= tol) {
                        pivot[i] = 1;
                        x[i] = (up[i] - xoffset[i]) / xscale[i];
                        newcon = true;
                    }
                }
            }
        }
        return newcon;
    }
    GD_HD bool removeConstraint(double gtpnew, double gnorm, double pgtolfs, double f, double fLastConstraint,
                                const double* g, int* pivot) const {
        if (((fLastConstraint - f) <= (gtpnew * -0.5)) && (gnorm > pgtolfs)) return false;
        int imax = -1;
        double cmax = 0.0;
        for (int i = 0; i < n; ++i) {
            if (pivot[i] == 2) continue;
            const double t = -pivot[i] * g[i];
            if (t < cmax) {
                cmax = t;
                imax = i;
            }
        }
        if (imax != -1) {
            pivot[imax] = 0;
            return true;
        }
        return false;
    }
    GD_HD static double initialStep(double fnew, double fmin, double gtp, double smax) {
        const double d = fabs(fnew - fmin);
        double alpha = 1.0;
        if (d * 2.0 <= -gtp && d >= EPSMCH) alpha = d * -2.0 / gtp;
        if (alpha >= smax) alpha = smax;
        return alpha;
    }
    GD_HD void ssbfgs(double gamma, const double* sj, const double* hjv, const double* hjyj, double yjsj, double yjhyj,
                      double vsj, double vhyj, double* hjp1v) const {
        double beta, delta;
        if (yjsj == 0.0) {
            delta = 0.0;
            beta = 0.0;
        } else {
            delta = (gamma * yjhyj / yjsj + 1.0) * vsj / yjsj - gamma * vhyj / yjsj;
            beta = -gamma * vsj / yjsj;
        }
        for (int i = 0; i < n; ++i) hjp1v[i] = gamma * hjv[i] + delta * sj[i] + beta * hjyj[i];
    }
    GD_HD void msolve(const double* g, double* y, const double* sk, const double* yk, const double* diagb,
                      const double* sr, const double* yr, bool upd1, double yksk, double yrsr, bool lreset) const {
        if (upd1) {
            for (int i = 0; i < n; ++i) y[i] = g[i] / diagb[i];
            return;
        }
        const double gsk = ddot1(g, sk);
        double hg[TNC_MAXN], hyk[TNC_MAXN], hyr[TNC_MAXN];
        if (lreset) {
            for (int i = 0; i < n; ++i) {
                const double rdiagb = 1.0 / diagb[i];
                hg[i] = g[i] * rdiagb;
                hyk[i] = yk[i] * rdiagb;
            }
            const double ykhyk = ddot1(yk, hyk);
            const double ghyk = ddot1(g, hyk);
            ssbfgs(1.0, sk, hg, hyk, yksk, ykhyk, gsk, ghyk, y);
        } else {
            for (int i = 0; i < n; ++i) {
                const double rdiagb = 1.0 / diagb[i];
                hg[i] = g[i] * rdiagb;
                hyk[i] = yk[i] * rdiagb;
                hyr[i] = yr[i] * rdiagb;
            }
            const double gsr = ddot1(g, sr);
            const double ghyr = ddot1(g, hyr);
            const double yrhyr = ddot1(yr, hyr);
            ssbfgs(1.0, sr, hg, hyr, yrsr, yrhyr, gsr, ghyr, hg);
            const double yksr = ddot1(yk, sr);
            const double ykhyr = ddot1(yk, hyr);
            ssbfgs(1.0, sr, hyk, hyr, yrsr, yrhyr, yksr, ykhyr, hyk);
            const double ykhyk = ddot1(hyk, yk);
            const double ghyk = ddot1(hyk, g);
            ssbfgs(1.0, sk, hg, hyk, yksk, ykhyk, gsk, ghyk, y);
        }
    }
    GD_HD void initPreconditioner(const double* diagb, double* emat, bool lreset, double yksk, double yrsr,
                                  const double* sk, const double* yk, const double* sr, const double* yr, bool upd1) const {
        if (upd1) {
            for (int i = 0; i < n; ++i) emat[i] = diagb[i];
            return;
        }
        double bsk[TNC_MAXN];
        if (lreset) {
            for (int i = 0; i < n; ++i) bsk[i] = diagb[i] * sk[i];
            double sds = ddot1(sk, bsk);
            if (yksk == 0.0) yksk = 1.0;
            if (sds == 0.0) sds = 1.0;
            for (int i = 0; i < n; ++i) {
                const double td = diagb[i];
                emat[i] = td - td * td * sk[i] * sk[i] / sds + yk[i] * yk[i] / yksk;
            }
        } else {
            for (int i = 0; i < n; ++i) bsk[i] = diagb[i] * sr[i];
            double sds = ddot1(sr, bsk);
            const double srds = ddot1(sk, bsk);
            const double yrsk = ddot1(yr, sk);
            if (yrsr == 0.0) yrsr = 1.0;
            if (sds == 0.0) sds = 1.0;
            for (int i = 0; i < n; ++i) {
                const double td = diagb[i];
                bsk[i] = td * sk[i] - bsk[i] * srds / sds + yr[i] * yrsk / yrsr;
                emat[i] = td - td * td * sr[i] * sr[i] / sds + yr[i] * yr[i] / yrsr;
            }
            sds = ddot1(sk, bsk);
            if (yksk == 0.0) yksk = 1.0;
            if (sds == 0.0) sds = 1.0;
            // sic: tnc.c SUBTRACTS the yk term here (Nash's Fortran adds it); scipy's results follow the C code
            for (int i = 0; i < n; ++i) emat[i] = emat[i] - (bsk[i] * bsk[i] / sds + yk[i] * yk[i] / yksk);
        }
    }
    GD_HD void diagonalScaling(double* e, const double* v, const double* gv, const double* r) const {
        const double vr = 1.0 / ddot1(v, r);
        const double vgv = 1.0 / ddot1(v, gv);
        for (int i = 0; i < n; ++i) {
            e[i] += -r[i] * r[i] * vr + gv[i] * gv[i] * vgv;
            if (e[i] <= 1e-6) e[i] = 1.0;
        }
    }
    GD_HD int hessianTimesVector(const double* v, double* gv, const double* x, const double* g, const double* xscale,
                                 const double* xoffset, double fscale, double accuracy, double xnorm) {
        double xv[TNC_MAXN], f;
        const double delta = accuracy * (xnorm + 1.0);
        for (int i = 0; i < n; ++i) xv[i] = x[i] + delta * v[i];
        unscalex(xv, xscale, xoffset);
        coercex(xv);
        if (function(xv, &f, gv)) return 1;
        scaleg(gv, xscale, fscale);
        const double dinv = 1.0 / delta;
        for (int i = 0; i < n; ++i) gv[i] = (gv[i] - g[i]) * dinv;
        for (int i = 0; i < n; ++i)
            if (xscale[i] == 0.0) gv[i] = 0.0;
        return 0;
    }
    GD_HD int direction(double* zsol, double* diagb, const double* x, const double* g, int maxCGit, int maxnfeval,
                        int* nfeval, bool upd1, double yksk, double yrsr, const double* sk, const double* yk,
                        const double* sr, const double* yr, bool lreset, const double* xscale, const double* xoffset,
                        double fscale, const int* pivot, double accuracy, double gnorm, double xnorm) {
        if (maxCGit == 0) {
            for (int i = 0; i < n; ++i) zsol[i] = -g[i];
            project(zsol, pivot);
            return 0;
        }
        const double rhsnrm = gnorm, tol = 1e-12;
        double qold = 0.0, rzold = 0.0;
        double r[TNC_MAXN], zk[TNC_MAXN], v[TNC_MAXN], emat[TNC_MAXN], gv[TNC_MAXN];
        initPreconditioner(diagb, emat, lreset, yksk, yrsr, sk, yk, sr, yr, upd1);
        for (int i = 0; i < n; ++i) {
            r[i] = -g[i];
            v[i] = 0.0;
            zsol[i] = 0.0;
        }
        int frc = 0;
        for (int k = 0; k < maxCGit; ++k) {
            project(r, pivot);
            GD_DBG("  msolve in: r=(%.17g,%.17g,%.17g) sk=(%.17g,%.17g,%.17g) yk=(%.17g,%.17g,%.17g) diagb=(%.17g,%.17g,%.17g) yksk=%.17g yrsr=%.17g sr=(%.17g,%.17g,%.17g) yr=(%.17g,%.17g,%.17g) xnorm=%.17g\n", r[0], r[1], r[2], sk[0], sk[1], sk[2], yk[0], yk[1], yk[2], diagb[0], diagb[1], diagb[2], yksk, yrsr, sr[0], sr[1], sr[2], yr[0], yr[1], yr[2], xnorm);
            msolve(r, zk, sk, yk, diagb, sr, yr, upd1, yksk, yrsr, lreset);
            project(zk, pivot);
            const double rz = ddot1(r, zk);
            if ((rz / rhsnrm < tol) || ((*nfeval) >= (maxnfeval - 1))) {
                if (dnrm21(zsol) == 0.0) {  // the preconditioner is not positive definite here: plain steepest descent
                    for (int i = 0; i < n; ++i) zsol[i] = -g[i];
                    project(zsol, pivot);
                }
                break;
            }
            const double beta = (k == 0) ? 0.0 : rz / rzold;
            for (int i = 0; i < n; ++i) v[i] = zk[i] + beta * v[i];
            project(v, pivot);
            frc = hessianTimesVector(v, gv, x, g, xscale, xoffset, fscale, accuracy, xnorm);
            ++(*nfeval);
            if (frc) return frc;
            project(gv, pivot);
            const double vgv = ddot1(v, gv);
            GD_DBG("  cg k=%d rz=%g vgv=%g rhsnrm=%g v=(%g,%g) gv=(%g,%g) zk=(%g,%g) g=(%g,%g)\n", k, rz, vgv, rhsnrm, v[0], v[1], gv[0], gv[1], zk[0], zk[1], g[0], g[1]);
            if (vgv / rhsnrm < tol) {
                if (dnrm21(zsol) == 0.0) {  // emergency exit before any progress: preconditioned steepest descent
                    msolve(g, zsol, sk, yk, diagb, sr, yr, upd1, yksk, yrsr, lreset);
                    for (int i = 0; i < n; ++i) zsol[i] = -zsol[i];
                    project(zsol, pivot);
                }
                break;
            }
            GD_DBG("  scaling in: emat=(%.17g,%.17g,%.17g) v=(%.17g,%.17g,%.17g) gv=(%.17g,%.17g,%.17g) r=(%.17g,%.17g,%.17g)\n", emat[0], emat[1], emat[2], v[0], v[1], v[2], gv[0], gv[1], gv[2], r[0], r[1], r[2]);
            diagonalScaling(emat, v, gv, r);
            GD_DBG("  scaling out: emat=(%.17g,%.17g,%.17g)\n", emat[0], emat[1], emat[2]);
            const double alpha = rz / vgv;
            for (int i = 0; i < n; ++i) zsol[i] += alpha * v[i];
            for (int i = 0; i < n; ++i) r[i] += -alpha * gv[i];
            const double gtp = ddot1(zsol, g);
            const double pr = ddot1(r, zsol);
            const double qnew = (gtp + pr) * 0.5;
            const double qtest = (k + 1) * (1.0 - qold / qnew);
            GD_DBG("  cg alpha=%g gtp=%g pr=%g qnew=%g qtest=%g zsol=(%g,%g)\n", alpha, gtp, pr, qnew, qtest, zsol[0], zsol[1]);
            if (qtest <= 0.5) break;
            if (gtp > 0.0) {
                for (int i = 0; i < n; ++i) zsol[i] += -alpha * v[i];
                break;
            }
            qold = qnew;
            rzold = rz;
        }
        for (int i = 0; i < n; ++i) diagb[i] = emat[i];
        return 0;
    }

    // --- Gill & Murray's getptc (safeguarded cubic interpolation step length)
    struct Ptc {
        double reltol, abstol, u, fu, gu, xmin, fmin, gmin, xw, fw, gw, a, b, oldf, b1, scxbnd, e, step, factor, gtest1,
            gtest2, tol;
        bool braktd;
    };
    enum { GETPTC_OK = 0, GETPTC_EVAL = 1, GETPTC_EINVAL = 2, GETPTC_FAIL = 3 };

    GD_HD static int getptcInit(Ptc& s, double tnytol, double eta, double rmu, double xbnd) {
        if (s.u <= 0.0 || xbnd <= tnytol || s.gu > 0.0) return GETPTC_EINVAL;
        if (xbnd < s.abstol) s.abstol = xbnd;
        s.tol = s.abstol;
        s.a = 0.0;
        s.xw = 0.0;
        s.xmin = 0.0;
        s.oldf = s.fu;
        s.fmin = s.fu;
        s.fw = s.fu;
        s.gw = s.gu;
        s.gmin = s.gu;
        s.step = s.u;
        s.factor = 5.0;
        s.braktd = false;
        s.scxbnd = xbnd;
        s.b = s.scxbnd + s.reltol * fabs(s.scxbnd) + s.abstol;
        s.e = s.b + s.b;
        s.b1 = s.b;
        s.gtest1 = -rmu * s.gu;
        s.gtest2 = -eta * s.gu;
        if (s.step >= s.scxbnd) {
            s.step = s.scxbnd;
            s.scxbnd -= (s.reltol * fabs(xbnd) + s.abstol) / (1.0 + s.reltol);
        }
        s.u = s.step;
        if (fabs(s.step) < s.tol && s.step < 0.0) s.u = -s.tol;
        if (fabs(s.step) < s.tol && s.step >= 0.0) s.u = s.tol;
        return GETPTC_EVAL;
    }

    GD_HD static int getptcIter(Ptc& s, double big, double rtsmll, double tnytol, double fpresn, double xbnd) {
        double abgw, absr, p, q, r, sv, scale, denom, a1, d1, d2, sumsq, abgmin, chordm, chordu, xmidpt, twotol;
        bool convrg;
        bool to_check = false;
        if (s.fu <= s.fmin) {
            chordu = s.oldf - (s.xmin + s.u) * s.gtest1;
            if (s.fu > chordu) {
                chordm = s.oldf - s.xmin * s.gtest1;
                s.gu = -s.gmin;
                denom = chordm - s.fmin;
                if (fabs(denom) < 1e-15) {
                    denom = 1e-15;
                    if (chordm - s.fmin < 0.0) denom = -denom;
                }
                if (s.xmin != 0.0) s.gu = s.gmin * (chordu - s.fu) / denom;
                s.fu = 0.5 * s.u * (s.gmin + s.gu) + s.fmin;
                if (s.fu < s.fmin) s.fu = s.fmin;
            } else {
                s.fw = s.fmin;
                s.fmin = s.fu;
                s.gw = s.gmin;
                s.gmin = s.gu;
                s.xmin += s.u;
                s.a -= s.u;
                s.b -= s.u;
                s.xw = -s.u;
                s.scxbnd -= s.u;
                if (s.gu <= 0.0) {
                    s.a = 0.0;
                } else {
                    s.b = 0.0;
                    s.braktd = true;
                }
                s.tol = fabs(s.xmin) * s.reltol + s.abstol;
                to_check = true;
            }
        }
        if (!to_check) {
            if (s.u < 0.0) {
                s.a = s.u;
            } else {
                s.b = s.u;
                s.braktd = true;
            }
            s.xw = s.u;
            s.fw = s.fu;
            s.gw = s.gu;
        }
        twotol = s.tol + s.tol;
        xmidpt = 0.5 * (s.a + s.b);
        convrg = (fabs(xmidpt) <= twotol - 0.5 * (s.b - s.a)) ||
                 (fabs(s.gmin) <= s.gtest2 && s.fmin < s.oldf && ((fabs(s.xmin - xbnd) > s.tol) || (!s.braktd)));
        if (convrg) {
            if (s.xmin != 0.0) return GETPTC_OK;
            if (fabs(s.oldf - s.fw) <= fpresn) return GETPTC_FAIL;
            s.tol = 0.1 * s.tol;
            if (s.tol < tnytol) return GETPTC_FAIL;
            s.reltol = 0.1 * s.reltol;
            s.abstol = 0.1 * s.abstol;
            twotol = 0.1 * twotol;
        }
        r = 0.0;
        q = 0.0;
        sv = 0.0;
        bool minimum_found = false;
        if (fabs(s.e) > s.tol) {
            r = 3.0 * (s.fmin - s.fw) / s.xw + s.gmin + s.gw;
            absr = fabs(r);
            q = absr;
            if (s.gw != 0.0 && s.gmin != 0.0) {
                abgw = fabs(s.gw);
                abgmin = fabs(s.gmin);
                sv = sqrt(abgmin) * sqrt(abgw);
                if (s.gw / abgw * s.gmin > 0.0) {
                    if (r >= sv || r <= -sv) {
                        q = sqrt(fabs(r + sv)) * sqrt(fabs(r - sv));
                    } else {
                        r = 0.0;
                        q = 0.0;
                        minimum_found = true;
                    }
                } else {
                    sumsq = 1.0;
                    p = 0.0;
                    if (absr >= sv) {
                        if (absr > rtsmll) p = absr * rtsmll;
                        if (sv >= p) {
                            const double value = sv / absr;
                            sumsq = 1.0 + value * value;
                        }
                        scale = absr;
                    } else {
                        if (sv > rtsmll) p = sv * rtsmll;
                        if (absr >= p) {
                            const double value = absr / sv;
                            sumsq = 1.0 + value * value;
                        }
                        scale = sv;
                    }
                    sumsq = sqrt(sumsq);
                    q = big;
                    if (scale < big / sumsq) q = scale * sumsq;
                }
            }
            if (!minimum_found) {
                if (s.xw < 0.0) q = -q;
                sv = s.xw * (s.gmin - r - q);
                q = s.gw - s.gmin + q + q;
                if (q > 0.0) sv = -sv;
                if (q <= 0.0) q = -q;
                r = s.e;
                if (s.b1 != s.step || s.braktd) s.e = s.step;
            }
        }
        a1 = s.a;
        s.b1 = s.b;
        s.step = xmidpt;
        if ((!s.braktd) || ((s.a == 0.0 && s.xw < 0.0) || (s.b == 0.0 && s.xw > 0.0))) {
            if (s.braktd) {
                d1 = s.xw;
                d2 = s.a;
                if (s.a == 0.0) d2 = s.b;
                s.u = -d1 / d2;
                s.step = 5.0 * d2 * (0.1 + 1.0 / s.u) / 11.0;
                if (s.u < 1.0) s.step = 0.5 * d2 * sqrt(s.u);
            } else {
                s.step = -s.factor * s.xw;
                if (s.step > s.scxbnd) s.step = s.scxbnd;
                if (s.step != s.scxbnd) s.factor = 5.0 * s.factor;
            }
            if (s.step <= 0.0) a1 = s.step;
            if (s.step > 0.0) s.b1 = s.step;
        }
        if (fabs(sv) <= fabs(0.5 * q * r) || sv <= q * a1 || sv >= q * s.b1) {
            s.e = s.b - s.a;
        } else {
            s.step = sv / q;
            if (s.step - s.a < twotol || s.b - s.step < twotol) {
                if (xmidpt <= 0.0)
                    s.step = -s.tol;
                else
                    s.step = s.tol;
            }
        }
        if (s.step >= s.scxbnd) {
            s.step = s.scxbnd;
            s.scxbnd -= (s.reltol * fabs(xbnd) + s.abstol) / (1.0 + s.reltol);
        }
        s.u = s.step;
        if (fabs(s.step) < s.tol && s.step < 0.0) s.u = -s.tol;
        if (fabs(s.step) < s.tol && s.step >= 0.0) s.u = s.tol;
        return GETPTC_EVAL;
    }

    enum { LS_OK = 0, LS_MAXFUN = 1, LS_FAIL = 2, LS_USERABORT = 3 };

    GD_HD int linearSearch(const double* xscale, const double* xoffset, double fscale, const int* pivot, double eta,
                           double ftol, double xbnd, const double* p, double* x, double* f, double* alpha, double* gfull,
                           int maxnfeval, int* nfeval) {
        double temp[TNC_MAXN], tempgfull[TNC_MAXN], newgfull[TNC_MAXN];
        const int maxlsit = 64;
        Ptc s;
        for (int i = 0; i < n; ++i) temp[i] = gfull[i], newgfull[i] = gfull[i];
        scaleg(temp, xscale, fscale);
        s.gu = ddot1(temp, p);
        for (int i = 0; i < n; ++i) temp[i] = x[i];
        project(temp, pivot);
        const double xnorm = dnrm21(temp);
        const double epsmch = EPSMCH, rteps = sqrt(epsmch);
        const double pe = dnrm21(p) + epsmch;
        s.reltol = rteps * (xnorm + 1.0) / pe;
        s.abstol = -epsmch * (1.0 + fabs(*f)) / (s.gu - epsmch);
        const double tnytol = epsmch * (xnorm + 1.0) / pe;
        const double rtsmll = epsmch, big = 1.0 / (epsmch * epsmch);
        int itcnt = 0;
        const double fpresn = ftol;
        s.u = *alpha;
        s.fu = *f;
        s.fmin = *f;
        const double rmu = 1e-4;
        s.xmin = *alpha;
        int itest = getptcInit(s, tnytol, eta, rmu, xbnd);
        *alpha = s.xmin;
        while (itest == GETPTC_EVAL) {
            if ((++itcnt > maxlsit) || ((*nfeval) >= maxnfeval)) break;
            const double ualpha = s.xmin + s.u;
            GD_DBG("  ls eval: xmin=%g u=%g ualpha=%g a=%g b=%g tol=%g\n", s.xmin, s.u, ualpha, s.a, s.b, s.tol);
            for (int i = 0; i < n; ++i) temp[i] = x[i] + ualpha * p[i];
            GD_DBG("  ls point: x=(%.17g,%.17g,%.17g) p=(%.17g,%.17g,%.17g) ualpha=%.17g temp=(%.17g,%.17g,%.17g) xmin=%.17g u=%.17g\n", x[0], x[1], x[2], p[0], p[1], p[2], ualpha, temp[0], temp[1], temp[2], s.xmin, s.u);
            unscalex(temp, xscale, xoffset);
            coercex(temp);
            const int frc = function(temp, &s.fu, tempgfull);
            ++(*nfeval);
            if (frc) return LS_USERABORT;
            s.fu *= fscale;
            for (int i = 0; i < n; ++i) temp[i] = tempgfull[i];
            scaleg(temp, xscale, fscale);
            s.gu = ddot1(temp, p);
            itest = getptcIter(s, big, rtsmll, tnytol, fpresn, xbnd);
            GD_DBG("  ls -> itest=%d fu=%g gu=%g fmin=%g\n", itest, s.fu, s.gu, s.fmin);
            if (s.xmin == ualpha)
                for (int i = 0; i < n; ++i) newgfull[i] = tempgfull[i];
        }
        *alpha = s.xmin;
        if (itest == GETPTC_OK) {
            *f = s.fmin;
            for (int i = 0; i < n; ++i) x[i] += *alpha * p[i];
            for (int i = 0; i < n; ++i) gfull[i] = newgfull[i];
            return LS_OK;
        }
        if (itcnt > maxlsit) return LS_FAIL;
        if (itest != GETPTC_EVAL) return LS_FAIL;
        return LS_MAXFUN;
    }

    GD_HD int minimize_scaled(double* x, double* f, double* gfull, const double* xscale, const double* xoffset,
                              double* fscale, int maxCGit, int maxnfeval, int* nfeval, int* niter, double eta, double stepmx,
                              double accuracy, double fmin, double ftol, double xtol, double pgtol, double rescale) {
        double difnew = 0.0, epsred = 0.05, oldgtp, difold, oldf, xnorm, newscale, gnorm, ustpmax, fLastConstraint, fLastReset,
               spe, yrsr = 0.0, yksk = 0.0, alpha = 0.0;
        double temp[TNC_MAXN], sk[TNC_MAXN], yk[TNC_MAXN], diagb[TNC_MAXN], sr[TNC_MAXN], yr[TNC_MAXN], oldg[TNC_MAXN],
            pk[TNC_MAXN], g[TNC_MAXN];
        int pivot[TNC_MAXN];
        const double epsmch = EPSMCH;
        bool upd1 = true, newcon = true, lreset = false, remcon;
        int icycle = n - 1, rc;
        *niter = 0;
        for (int i = 0; i < n; ++i) sk[i] = yk[i] = sr[i] = yr[i] = 0.0;
        scalex(x, xscale, xoffset);
        (*f) *= *fscale;
        setConstraints(x, pivot, xscale, xoffset);
        for (int i = 0; i < n; ++i) g[i] = gfull[i];
        scaleg(g, xscale, *fscale);
        for (int i = 0; i < n; ++i)
            if (-pivot[i] * g[i] < 0.0) pivot[i] = 0;
        project(g, pivot);
        gnorm = dnrm21(g);
        fLastConstraint = *f;
        fLastReset = *f;
        for (int i = 0; i < n; ++i) diagb[i] = 1.0;
        for (;;) {
            if (dnrm21(g) <= pgtol * (*fscale)) {
                rc = TNC_LOCALMINIMUM;
                break;
            }
            if (*nfeval >= maxnfeval) {
                rc = TNC_MAXFUN;
                break;
            }
            newscale = dnrm21(g);
            if ((newscale > epsmch) && (fabs(log10(newscale)) > rescale)) {
                newscale = 1.0 / newscale;
                *f *= newscale;
                *fscale *= newscale;
                gnorm *= newscale;
                fLastConstraint *= newscale;
                fLastReset *= newscale;
                difnew *= newscale;
                for (int i = 0; i < n; ++i) g[i] *= newscale;
                for (int i = 0; i < n; ++i) diagb[i] = 1.0;
                upd1 = true;
                icycle = n - 1;
                newcon = true;
            }
            for (int i = 0; i < n; ++i) temp[i] = x[i];
            project(temp, pivot);
            xnorm = dnrm21(temp);
            const int oldnfeval = *nfeval;
            const int frc = direction(pk, diagb, x, g, maxCGit, maxnfeval, nfeval, upd1, yksk, yrsr, sk, yk, sr, yr, lreset,
                                      xscale, xoffset, *fscale, pivot, accuracy, gnorm, xnorm);
            if (frc) {
                rc = TNC_USERABORT;
                break;
            }
            if (!newcon) {
                if (!lreset) {
                    for (int i = 0; i < n; ++i) sr[i] += sk[i];
                    for (int i = 0; i < n; ++i) yr[i] += yk[i];
                    icycle++;
                } else {
                    for (int i = 0; i < n; ++i) sr[i] = sk[i];
                    for (int i = 0; i < n; ++i) yr[i] = yk[i];
                    fLastReset = *f;
                    icycle = 1;
                }
            }
            for (int i = 0; i < n; ++i) oldg[i] = g[i];
            oldf = *f;
            oldgtp = ddot1(pk, g);
            ustpmax = stepmx / (dnrm21(pk) + epsmch);
            spe = stepMax(ustpmax, x, pk, pivot, xscale, xoffset);
            GD_DBG("iter %d: pk=(%g,%g,%g) oldgtp=%g ustpmax=%g spe=%g f=%g fscale=%g upd1=%d lreset=%d newcon=%d\n", *niter, pk[0], pk[1], n > 2 ? pk[2] : 0.0, oldgtp, ustpmax, spe, *f, *fscale, (int)upd1, (int)lreset, (int)newcon);
            if (spe > 0.0) {
                alpha = initialStep(*f, fmin / (*fscale), oldgtp, spe);
                GD_DBG("  initial alpha=%g\n", alpha);
                const int lsrc = linearSearch(xscale, xoffset, *fscale, pivot, eta, ftol, spe, pk, x, f, &alpha, gfull,
                                              maxnfeval, nfeval);
                if (lsrc == LS_USERABORT) {
                    rc = TNC_USERABORT;
                    break;
                }
                if (lsrc == LS_FAIL) {
                    rc = TNC_LSFAIL;
                    break;
                }
                if (alpha >= 0.9 * ustpmax) stepmx *= 1e2;
                if (alpha - spe >= -epsmch * 10.0) {
                    newcon = true;
                } else {
                    if (lsrc != LS_OK) {
                        rc = (lsrc == LS_MAXFUN) ? TNC_MAXFUN : TNC_LSFAIL;
                        break;
                    }
                    newcon = false;
                }
            } else {
                newcon = true;
            }
            if (newcon) {
                if (!addConstraint(x, pk, pivot, xscale, xoffset)) {
                    if (*nfeval == oldnfeval) {
                        rc = TNC_NOPROGRESS;
                        break;
                    }
                }
                fLastConstraint = *f;
            }
            (*niter)++;
            difold = difnew;
            difnew = oldf - *f;
            if (icycle == 1) {
                if (difnew > difold * 2.0) epsred += epsred;
                if (difnew < difold * 0.5) epsred *= 0.5;
            }
            for (int i = 0; i < n; ++i) g[i] = gfull[i];
            scaleg(g, xscale, *fscale);
            for (int i = 0; i < n; ++i) temp[i] = g[i];
            project(temp, pivot);
            gnorm = dnrm21(temp);
            remcon = removeConstraint(oldgtp, gnorm, pgtol * (*fscale), *f, fLastConstraint, g, pivot);
            if (remcon) {
                for (int i = 0; i < n; ++i) temp[i] = g[i];
                project(temp, pivot);
                gnorm = dnrm21(temp);
                fLastConstraint = *f;
            }
            if (!remcon && !newcon) {
                if (fabs(difnew) <= ftol * (*fscale)) {
                    rc = TNC_FCONVERGED;
                    break;
                }
                if (alpha * dnrm21(pk) <= xtol) {
                    rc = TNC_XCONVERGED;
                    break;
                }
            }
            project(g, pivot);
            if (!newcon) {  // a released constraint does not suppress the quasi-Newton update, a new one does
                for (int i = 0; i < n; ++i) {
                    yk[i] = g[i] - oldg[i];
                    sk[i] = alpha * pk[i];
                }
                yksk = ddot1(yk, sk);
                if (icycle == (n - 1) || difnew < epsred * (fLastReset - *f)) {
                    lreset = true;
                } else {
                    yrsr = ddot1(yr, sr);
                    lreset = (yrsr <= 0.0);
                }
                upd1 = false;
            }
        }
        unscalex(x, xscale, xoffset);
        coercex(x);
        (*f) /= *fscale;
        return rc;
    }

    // tnc() with scipy's arguments: scale = offset = NULL, maxCGit = eta = ftol = xtol = pgtol = rescale = -1,
    // stepmx = accuracy = fmin = 0, maxnfeval = max(100, 10 n)
    GD_HD TncResult run(const double* x0, const double* lo, const double* hi) {
        TncResult res;
        for (int i = 0; i < n; ++i) low[i] = lo[i], up[i] = hi[i], res.x[i] = x0[i];
        res.rc = TNC_USERABORT;
        res.nfev = 0;
        res.niter = 0;
        res.success = false;
        double x[TNC_MAXN], f, g[TNC_MAXN];
        for (int i = 0; i < n; ++i) x[i] = x0[i];
        // ScalarFunction.__init__ evaluates f and the gradient at the UNCLIPPED x0 (raising if it violates the bounds)
        if (function(x, &f, g)) return res;
        int nfeval = 0, niter = 0;
        const int maxnfeval = (10 * n > 100) ? 10 * n : 100;
        for (int i = 0; i < n; ++i)
            if (low[i] > up[i]) {
                res.rc = TNC_INFEASIBLE;
                return res;
            }
        coercex(x);
        if (function(x, &f, g)) return res;
        nfeval++;
        int nc = 0;
        for (int i = 0; i < n; ++i)
            if (low[i] == up[i]) nc++;
        if (nc == n) {
            res.rc = TNC_CONSTANT;
            return res;
        }
        double xscale[TNC_MAXN], xoffset[TNC_MAXN], fscale = 1.0;
        for (int i = 0; i < n; ++i) {
            if (low[i] != -TNC_HUGE && up[i] != TNC_HUGE) {
                xscale[i] = up[i] - low[i];
                xoffset[i] = (up[i] + low[i]) * 0.5;
            } else {
                xscale[i] = 1.0 + fabs(x[i]);
                xoffset[i] = x[i];
            }
        }
        const double rteps = sqrt(EPSMCH);
        double stepmx = 0.0, eta = -1.0, rescale = -1.0, accuracy = 0.0, ftol = -1.0, pgtol = -1.0, xtol = -1.0;
        int maxCGit = -1;
        if (stepmx < rteps * 10.0) stepmx = 1.0e1;
        if (eta < 0.0 || eta >= 1.0) eta = 0.25;
        if (rescale < 0) rescale = 1.3;
        if (maxCGit < 0) {
            maxCGit = n / 2;
            if (maxCGit < 1)
                maxCGit = 1;
            else if (maxCGit > 50)
                maxCGit = 50;
        }
        if (maxCGit > n) maxCGit = n;
        if (accuracy <= EPSMCH) accuracy = rteps;
        if (ftol < 0.0) ftol = accuracy;
        if (pgtol < 0.0) pgtol = 1e-2 * sqrt(accuracy);
        if (xtol < 0.0) xtol = rteps;
        res.rc = minimize_scaled(x, &f, g, xscale, xoffset, &fscale, maxCGit, maxnfeval, &nfeval, &niter, eta, stepmx,
                                 accuracy, 0.0, ftol, xtol, pgtol, rescale);
        for (int i = 0; i < n; ++i) res.x[i] = x[i];
        res.nfev = nfeval;
        res.niter = niter;
        res.success = (res.rc > -1 && res.rc < 3);
        return res;
    }
};


// ---------------------------------------------------------------------------------------------------------------
// KernelOptimizer2D.get_h (kde_bandwidth.py:216-306): closed-form axis bandwidths from the psi functionals, then --
// for pairs without hard limits -- the correlated kernel by minimising the AMISE with TNC, first over (hx, hy) at the
// sample correlation, then over (hx, hy, c), each result accepted only if it lowers the AMISE (by 10 % for the second).
// psi = (psi_02, psi_20, psi_11, psi_00, psi_13, psi_31); bandwidths in units of the bin range.
// numpy / Python evaluate x**2 as libm pow(x, 2.0), which is NOT always the correctly rounded product x*x (0.08 % of
// arguments differ by an ulp in glibc 2.35).  On the host (the test harness, compared bit for bit with scipy) call the
// real pow through an opaque exponent so that the compiler cannot fold it; on the device there is no glibc to match.
GD_HD inline double square_like_libm(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    return x * x;
#else
    volatile double two = 2.0;
    return pow(x, two);
#endif
}

struct Amise {
    double p40, p04, p22, p13, p31, N, corr;
    bool fixed_corr;  // the reference's AMISE(cov, corr): corr given -> c = corr, else c = cov[2]
    GD_HD double operator()(const double* cov, bool* fail) const {
        const double PI_ = 3.141592653589793;
        const double hx = cov[0], hy = cov[1];
        const double c = fixed_corr ? corr : cov[2];
        const double c2 = square_like_libm(c);
        const double var = 1.0 / (4 * PI_ * hx * hy * sqrt(1 - c2) * N);
        const double hx2 = square_like_libm(hx), hy2 = square_like_libm(hy);
        const double bias = 0.25 * (pow(hx, 4.0) * p40 + pow(hy, 4.0) * p04 + 2 * hx2 * hy2 * p22 * (2 * c2 + 1) +
                                    4 * c * hx * hy * (hx2 * p31 + hy2 * p13));
        if (bias < 0) {  // "bias not positive definite"
            *fail = true;
            return 0.0;
        }
        return var + bias;
    }
};

struct GetHResult {
    double hx, hy, corr;
    int status;  // 0 ok; 1 = the AMISE at the closed-form bandwidths is not positive definite (the reference raises)
    int nfev;
};

GD_HD inline GetHResult get_h(const double* psi, double N, double corr_in, bool do_correlation) {
    const double PI_ = 3.141592653589793;
    const double p_02 = psi[0], p_20 = psi[1], p_11 = psi[2];
    GetHResult out;
    out.status = 0;
    out.nfev = 0;
    double h_x = pow(pow(p_02, 3.0 / 4) / (4 * PI_ * N * pow(p_20, 3.0 / 4) * (p_11 + sqrt(p_20 * p_02))), 1.0 / 6);
    double h_y = pow(pow(p_20, 3.0 / 4) / (4 * PI_ * N * pow(p_02, 3.0 / 4) * (p_11 + sqrt(p_20 * p_02))), 1.0 / 6);
    double corr = 0;
    out.hx = h_x, out.hy = h_y, out.corr = corr;
    if (!do_correlation) return out;
    Amise am;
    am.p04 = p_02, am.p40 = p_20, am.p22 = p_11, am.p13 = psi[4], am.p31 = psi[5], am.N = N;
    am.corr = 0.0;
    am.fixed_corr = false;
    bool fail = false;
    const double start[3] = {h_x, h_y, 0.0};
    double AMISE = am(start, &fail);
    if (fail) {
        out.status = 1;
        return out;
    }
    if (corr_in != 0.0) {
        Amise fixed = am;
        fixed.fixed_corr = true;
        fixed.corr = corr_in;
        const double sc = sqrt(1 - fabs(corr_in));
        const double x0[2] = {h_x / sc, h_y / sc};
        const double lo[2] = {0.001, 0.001}, hi[2] = {0.3, 0.3};
        Tnc<Amise> tnc(fixed, 2);
        const TncResult r = tnc.run(x0, lo, hi);
        out.nfev += tnc.nfev_total;
        if (r.success) {
            bool f2 = false;
            const double a = fixed(r.x, &f2);
            if (!f2 && a < AMISE) {
                h_x = r.x[0], h_y = r.x[1];
                corr = corr_in;
                AMISE = a;
            }
        }
    }
    {
        const double x0[3] = {h_x, h_y, corr_in};
        const double lo[3] = {0.001, 0.001, -0.99}, hi[3] = {0.3, 0.3, 0.99};
        Tnc<Amise> tnc(am, 3);
        const TncResult r = tnc.run(x0, lo, hi);
        out.nfev += tnc.nfev_total;
        if (r.success) {
            bool f2 = false;
            const double a = am(r.x, &f2);
            if (!f2 && a < AMISE * 0.9) h_x = r.x[0], h_y = r.x[1], corr = r.x[2];
        }
    }
    out.hx = h_x, out.hy = h_y, out.corr = corr;
    return out;
}

}  // namespace gdsolve
