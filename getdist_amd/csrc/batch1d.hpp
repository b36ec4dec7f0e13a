// gd_density1d_batch: MCSamples.get1DDensityGridData (mcsamples.py:1500-1686) for B parameters in one call -- the host
// decisions of the reference (bin edges, the scalar tail of getAutoBandwidth1D, the smoothing scale in fine-bin units, the
// window half-width) made here, the histograms kept in device memory from the binning to the finished densities.
// Host C++ only, like batch2d.hpp, and shared with the CPU test harness (tests/native/batch_harness.cpp).
#pragma once

#include "batch2d.hpp"

namespace gdb {

struct Ops1D {
    // gd_hist1d_dev / gd_isj1d_dev / gd_density1d_dev of include/gdhip.h, handle first
    int (*hist1d_dev)(void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                      void* d_hist);
    int (*isj1d_dev)(void* h, int32_t B, int32_t F, const void* d_hist, const double* neff, double* hfrac, int32_t* status);
    int (*density1d_dev)(void* h, int32_t B, int32_t F, const void* d_hist, const double* smooth, const int32_t* winw,
                         const int32_t* flags, int32_t bco, int32_t mbc, double* P_out, int32_t* status);
    // blocking copy of device memory to the host, ordered behind the kernels above
    int (*fetch)(void* h, void* dst, const void* d_src, int64_t bytes);
};

struct Smoothing1D {
    double kde_h = NAN, smooth = NAN, neff = NAN;
    double h_isj = NAN;  // the solver's width when the fallback replaced it (NaN: the solver returned None): the message's "h="
    int32_t winw = 0, flags = 0, bits = 0;
};

// Everything between the ISJ solution and the convolution for one parameter: the scalar tail of getAutoBandwidth1D
// (mcsamples.py:1256-1283: rule-of-thumb fallback when the solver failed or the width is very small, the higher-order
// rescaling) and the smoothing scale of get1DDensityGridData (mcsamples.py:1563-1586).  `have_h` false = the solver returned
// None.  Returns false when the fallback was needed and settings.raise_on_bandwidth_errors is set.
static inline bool smoothing_1d(const gd_density1d_settings& s, const gd_param2d& p, double binmin, double binmax, bool have_h,
                                double h, Smoothing1D* out) {
    const int F = s.fine_bins;
    const double fine_width = (binmax - binmin) / (double)(F - 1);
    const double paramrange = p.range_max - p.range_min;
    double smooth;
    if (s.smooth_scale_1D <= 0) {
        const double N_eff = p.neff;
        out->neff = N_eff;
        if (!have_h) out->bits |= 1;
        const double bin_range = np_maximum(p.param_max, p.range_max) - np_minimum(p.param_min, p.range_min);
        if (!have_h || h < 0.01 * py_pow(N_eff, -1.0 / 5) * (p.range_max - p.range_min) / bin_range) {
            // (p.owned & 2): the parameter is in no_warning_params / a chi2 parameter under no_warning_chi2_params: the
            // reference then neither warns nor raises (mcsamples.py:1259-1266) and takes the fallback silently
            const bool quiet = (p.owned & 2) != 0;
            if (!quiet) out->bits |= 2;
            out->h_isj = have_h ? h : NAN;  // what the message quotes as h=
            h = 1.06 * p.sigma_range * py_pow(N_eff, -1.0 / 5) / bin_range;
            out->kde_h = h;  // ... and as "Using fallback (h=...)"
            if (s.raise_on_bandwidth_errors && !quiet) return false;
        }
        out->kde_h = h;
        int m = s.mult_bias_correction_order;
        if (s.boundary_correction_order > 1 && m < 1) m = 1;
        double bandwidth = m ? h * py_pow(N_eff, 1.0 / 5 - 1.0 / (double)(4 * m + 5)) : h;
        bandwidth = bandwidth * (binmax - binmin);
        if (paramrange / 4 < bandwidth) bandwidth = paramrange / 4;
        smooth = bandwidth * fabs(s.smooth_scale_1D) / fine_width;
    } else if (s.smooth_scale_1D < 1.0) {
        smooth = s.smooth_scale_1D * p.err / fine_width;
    } else {
        const double width = paramrange / (double)(s.num_bins - 1);
        smooth = s.smooth_scale_1D * width / fine_width;
    }
    if (smooth < 2) out->bits |= 4;  // "fine_bins not large enough to well sample smoothing scale"
    if (!(smooth > 1.0)) smooth = 1.0;          // max(1.0, smooth): a NaN stays out, as Python's max keeps its first argument
    if ((double)(F / 2) < smooth) smooth = F / 2;
    out->smooth = smooth;
    const int64_t lim = (int64_t)((p.periodic ? F - 1 : F) / 2 - 2);
    const int64_t w = (int64_t)nearbyint(2.5 * smooth);  // int(round(.)): ties to even
    out->winw = (int32_t)(w < lim ? w : lim);
    out->flags = (p.has_limits_bot ? 1 : 0) | (p.has_limits_top ? 2 : 0) | (p.periodic ? 4 : 0);
    return true;
}

// meta[b]: GD_BATCH1D_META doubles, documented in gdhip.h
static inline int density1d_batch(State& st, const Ops& ops, const Ops1D& o1, void* h, const gd_density1d_settings& s,
                                  gd_param2d* par, int n, const int32_t* cols, int B, double* P_out, double* hist_out,
                                  double* meta, std::string* err) {
    auto fail = [&](int code, const std::string& m) {
        if (err && err->empty()) *err = m;
        return code;
    };
    auto dev_fail = [&](int code) {
        const char* m = ops.last_error(h);
        return fail(code, m && *m ? m : "device call failed");
    };
    char buf[256];
    const int F = s.fine_bins;
    if (F < 8 || F > 4096) return fail(GD_ERR_BADARG, "fine_bins out of range (8..4096)");
    if (s.boundary_correction_order > 2) return fail(GD_ERR_BADARG, "Unknown boundary_correction_order (expected 0, 1, 2)");
    if (s.smooth_scale_1D >= 1.0 && s.num_bins < 2) return fail(GD_ERR_BADARG, "num_bins must be at least 2");
    std::vector<double> binmin(B), binmax(B), width(B);
    for (int b = 0; b < B; ++b) {
        if (cols[b] < 0 || cols[b] >= n) return fail(GD_ERR_BADARG, "column out of range");
        const gd_param2d& p = par[cols[b]];
        if (p.range_max - p.range_min <= 0) {
            snprintf(buf, sizeof buf, "Parameter range is <= 0: column %d", (int)cols[b]);
            return fail(GD_ERR_BADARG, buf);
        }
        bin_edges(p, &binmin[b], &binmax[b]);
        width[b] = (binmax[b] - binmin[b]) / (double)(F - 1);
    }
    // the histograms' block comes from the context's pool (shared with the 2D entry) and goes back to it on every way out;
    // a way out with kernels still in flight (an error after the binning was enqueued) waits for the stream first
    Pool pool{st, ops, h};
    int rc = 0;
    void* d_hist = pool.take((int64_t)B * F * 8, &rc);
    if (!d_hist) return dev_fail(rc);
    struct Release {
        Pool& pool;
        const Ops1D& o1;
        void* h;
        void* p;
        bool idle;
        ~Release() {
            double word;
            if (!idle) o1.fetch(h, &word, p, 8);
            pool.give(p);
        }
    } release{pool, o1, h, d_hist, false};
    rc = o1.hist1d_dev(h, cols, B, binmin.data(), width.data(), F, d_hist);
    if (rc) return dev_fail(rc);
    std::vector<double> hfrac(B, NAN);
    std::vector<int32_t> isj_status(B, 0);
    const bool automatic = s.smooth_scale_1D <= 0;
    if (automatic) {
        // effective sample numbers of the parameters that have none yet (_get1DNeff, mcsamples.py:1230-1235): the batched
        // route of the 2D entry -- its kernels run behind the binning on the same stream
        bool need = false;
        for (int b = 0; b < B; ++b) need = need || isnan(par[cols[b]].neff);
        if (need) {
            gd_batch2d_settings s2{};
            s2.norm = s.norm, s2.sum_w2 = s.sum_w2, s2.uncorrelated_sampler = s.uncorrelated_sampler;
            Call call(st, ops, h, nullptr, s2, par, n, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0,
                      nullptr, nullptr, nullptr, nullptr);
            int64_t ncols = 0;
            rc = ops.num_rows(h, &call.N, &ncols);
            if (rc) return dev_fail(rc);
            std::vector<int> js;
            for (int b = 0; b < B; ++b)
                if (std::find(js.begin(), js.end(), (int)cols[b]) == js.end()) js.push_back(cols[b]);
            rc = call.neff_batch(js, false);
            if (rc) return fail(rc, call.err);
        }
        std::vector<double> neff(B);
        for (int b = 0; b < B; ++b) {
            neff[b] = par[cols[b]].neff;
            if (!(neff[b] > 0)) {
                snprintf(buf, sizeof buf, "effective sample number of column %d is not positive", (int)cols[b]);
                return fail(GD_ERR_BADARG, buf);
            }
        }
        rc = o1.isj1d_dev(h, B, F, d_hist, neff.data(), hfrac.data(), isj_status.data());
        if (rc) return dev_fail(rc);
    }
    std::vector<double> smooth(B);
    std::vector<int32_t> winw(B), flags(B);
    for (int b = 0; b < B; ++b) {
        const gd_param2d& p = par[cols[b]];
        Smoothing1D sm;
        if (!smoothing_1d(s, p, binmin[b], binmax[b], isj_status[b] == 0, hfrac[b], &sm)) {
            // the reference's message (mcsamples.py:1262); the binding prints the three numbers as Python's repr does
            if (isj_status[b] != 0)
                snprintf(buf, sizeof buf, "auto bandwidth for column %d very small or failed (h=None,N_eff=%.17g). Using fallback (h=%.17g)",
                         (int)cols[b], sm.neff, sm.kde_h);
            else
                snprintf(buf, sizeof buf, "auto bandwidth for column %d very small or failed (h=%.17g,N_eff=%.17g). Using fallback (h=%.17g)",
                         (int)cols[b], sm.h_isj, sm.neff, sm.kde_h);
            return fail(GD_ERR_SOLVER, buf);
        }
        smooth[b] = sm.smooth, winw[b] = sm.winw, flags[b] = sm.flags;
        double* m = meta + (size_t)b * GD_BATCH1D_META;
        m[0] = binmin[b], m[1] = binmax[b], m[2] = sm.kde_h, m[3] = sm.smooth, m[4] = sm.winw, m[5] = sm.bits, m[6] = sm.neff;
        m[7] = 0, m[8] = sm.h_isj;
        if (!(sm.smooth > 0) || sm.winw < 0 || 2 * sm.winw + 1 > F) {
            snprintf(buf, sizeof buf, "smoothing scale of column %d is not usable (smooth_1D=%g)", (int)cols[b], sm.smooth);
            return fail(GD_ERR_BADARG, buf);
        }
    }
    std::vector<int32_t> status(B, 0);
    rc = o1.density1d_dev(h, B, F, d_hist, smooth.data(), winw.data(), flags.data(), s.boundary_correction_order,
                          s.mult_bias_correction_order, P_out, status.data());
    if (rc) return dev_fail(rc);
    release.idle = true;  // the densities are on the host: the stream has drained
    for (int b = 0; b < B; ++b) meta[(size_t)b * GD_BATCH1D_META + 7] = status[b];
    if (hist_out) {
        rc = o1.fetch(h, hist_out, d_hist, (int64_t)B * F * 8);
        if (rc) return dev_fail(rc);
    }
    return 0;
}

}  // namespace gdb
