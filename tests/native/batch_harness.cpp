// Host-only build (g++) of getdist_amd/csrc/batch2d.hpp for the CPU test-suite: the plan and the choreography of
// gd_density2d_batch run here exactly as inside libgdhip.so, with the table of device entry points supplied by the
// caller (tests/native_batch_util.py binds it to the numpy context double).  Test infrastructure only.
#include "../../getdist_amd/csrc/batch1d.hpp"
#include "../../getdist_amd/csrc/batch2d.hpp"

extern "C" {

void* gdt_batch_state_new() { return new gdb::State(); }

void gdt_batch_state_free(const gdb::Ops* ops, void* st, void* h) {
    gdb::State* s = (gdb::State*)st;
    gdb::release_all(*s, *ops, h);
    delete s;
}

int gdt_batch_finish(const gdb::Ops* ops, void* st, void* h) { return gdb::finish_all(*(gdb::State*)st, *ops, h); }

void gdt_batch_invalidate(void* st) { gdb::invalidate_index_columns(*(gdb::State*)st); }

int64_t gdt_batch_exchanges(void* st) {
    gdb::State* s = (gdb::State*)st;
    std::lock_guard<std::mutex> g(s->mu);
    return s->exchanges_entered;
}

int gdt_density2d_batch(const gdb::Ops* ops, void* st, void* h, void* twin, const gd_batch2d_settings* settings,
                        gd_param2d* params, int32_t n, const double* corr, const double* cov, const double* lag_probe,
                        const int32_t* pairs, int32_t P, gd_neff_exchange_fn exchange, void* exchange_user, double* grids,
                        int64_t grids_doubles, int32_t* status, double* meta, double* levels, int32_t* level_status,
                        int32_t* tokens_out2, char* errbuf, int32_t errlen) {
    int rc;
    try {
        gdb::Call call(*(gdb::State*)st, *ops, h, twin, *settings, params, n, corr, cov, lag_probe, pairs, P, exchange,
                       exchange_user, grids, grids_doubles, status, meta, levels, level_status);
        rc = call.run(tokens_out2);
        if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "%s", call.err.c_str());
    } catch (const std::exception& e) {
        rc = -99;
        if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "exception: %s", e.what());
    }
    return rc;
}

int gdt_density1d_batch(const gdb::Ops* ops, const gdb::Ops1D* ops1, void* st, void* h, const gd_density1d_settings* settings,
                        gd_param2d* params, int32_t n, const int32_t* cols, int32_t B, double* P_out, double* hist_out,
                        double* meta, char* errbuf, int32_t errlen) {
    int rc;
    try {
        std::string err;
        rc = gdb::density1d_batch(*(gdb::State*)st, *ops, *ops1, h, *settings, params, n, cols, B, P_out, hist_out, meta, &err);
        if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "%s", err.c_str());
    } catch (const std::exception& e) {
        rc = -99;
        if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "exception: %s", e.what());
    }
    return rc;
}

// the scalar tail alone: out8 = {kde_h, smooth, neff, winw, flags, bits, ok, 0}
void gdt_smoothing_1d(const gd_density1d_settings* s, const gd_param2d* p, double binmin, double binmax, int have_h, double h,
                      double* out8) {
    gdb::Smoothing1D sm;
    const bool ok = gdb::smoothing_1d(*s, *p, binmin, binmax, have_h != 0, h, &sm);
    out8[0] = sm.kde_h, out8[1] = sm.smooth, out8[2] = sm.neff, out8[3] = sm.winw, out8[4] = sm.flags, out8[5] = sm.bits;
    out8[6] = ok ? 1 : 0, out8[7] = 0;
}

int gdt_grid_sizes(const gd_batch2d_settings* settings, int32_t n, const double* corr, const int32_t* pairs, int32_t P,
                   int32_t* F_out) {
    gdb::PairScalars ps;
    gdb::pair_scalars(*settings, n, corr, pairs, P, ps);
    for (int k = 0; k < P; ++k) F_out[k] = ps.F[k];
    return 0;
}

int gdt_chol_shear(double c00, double c10, double c11, double* S4, double* r2) { return gdb::chol_shear(c00, c10, c11, S4, r2) ? 0 : 1; }

int gdt_frame_size(int n) { return gdb::frame_size(n); }

double gdt_py_pow(double x, double y) { return gdb::py_pow(x, y); }
}
