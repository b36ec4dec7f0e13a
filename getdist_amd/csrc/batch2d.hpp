// batch2d.hpp -- the host side of gd_density2d_batch: every decision MCSamples.get2DDensityGridData and
// getAutoBandwidth2D make for a batch of parameter pairs (/root/reference/getdist/mcsamples.py:1285-1419, 1486-1498,
// 1748-2010; chains.py:477-574 for the effective sample numbers), in the reference's own expression order, plus the
// choreography of the device work over two streams.
//
// Pure C++17, no HIP: the device is reached through a table of entry points (`Ops`).  libgdhip.so binds the table to its
// own C ABI (batch2d.hip); the CPU test-suite binds it to callbacks of the numpy context double and checks the grids of
// this orchestration against the Python-planned path (tests/native/batch_harness.cpp, tests/test_native_batch.py).
//
// Scalars that decide bits downstream (bin edges, shear coefficients, fallback times, window sizes) are evaluated with
// exactly the operations numpy / CPython perform: libm `pow` where Python writes `**` on scalars, x*x where numpy
// squares an array, round-half-even where numpy rounds, the LAPACK 2 x 2 factorisations spelled out (`chol_shear`).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gdhip.h"

namespace gdb {

// ---- device entry points (the C ABI of include/gdhip.h, handle first) -------------------------------------------------
struct Ops {
    int (*bind_thread)(void* h);
    int (*num_rows)(void* h, int64_t* N, int64_t* n);
    int (*weights_kind)(void* h, int32_t* has_weights);  // 0: unit weights (chains.py:313-315), 1: weights, 2: real weights
                                                         //    whose byte-index binning the library serves (sorted by stripe)
    int (*dev_alloc)(void* h, int64_t bytes, void** out);
    int (*dev_free)(void* h, void* p);
    int (*prebin8_batch)(void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                         void* const* d_idx, int64_t* bad);
    int (*hist2d_prebinned8)(void* h, int32_t B, const void* const* ix, const void* const* iy, void* d_hist);
    // both of the above back to back with one wait (ncols may be 0); GD_ERR_SOLVER: a sample outside the grid or a wrapped counter
    int (*prebin8_hist2d)(void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width,
                          void* const* d_idx, int64_t* bad, int32_t B, const void* const* ix, const void* const* iy, void* d_hist);
    int (*prebin)(void* h, int32_t col, double binmin, double width, int32_t F, void* d_idx);
    int (*hist2d_prebinned)(void* h, int32_t B, const void* const* ix, const void* const* iy, int32_t F, void* d_hist);
    int (*minmax_affine)(void* h, int32_t B, const int32_t* ci, const int32_t* cj, const double* a, const double* b, double* out);
    int (*hist2d_sheared)(void* h, int32_t B, const int32_t* ci, const int32_t* cj, const double* r0, const double* r1,
                          const double* xmin, const double* dx, const double* ymin, const double* dy, int32_t F, void* d_hist);
    int (*kopt2d)(void* h, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr,
                  const double* fallback_t, const double* corr, double* out);
    // d_dst[(dst_first + k)] = d_src[index[k]], items of item_bytes
    int (*gather_items)(void* h, void* d_dst, int64_t dst_first, const void* d_src, const int32_t* index, int32_t count,
                        int64_t item_bytes);
    // hist_index (may be null: the batch is d_hist's first B grids): pair b convolves histogram hist_index[b] of d_hist
    int (*density2d_enqueue)(void* h, int32_t B, int32_t F, const void* d_hist, const int32_t* hist_index, const double* rx,
                             const double* ry, const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco,
                             int32_t mbc, void* d_P, int32_t* status_pinned);
    int (*d2h_async)(void* h, void* dst, const void* d_src, int64_t bytes);
    int (*copy_mark)(void* h, int32_t* token);
    int (*copy_wait)(void* h, int32_t token);
    int (*copy_sync)(void* h);
    int (*contour_levels)(void* h, int32_t B, int32_t F, const void* d_P, const double* contours, int32_t nc, double* out,
                          int32_t* status);
    int (*autocov_lags_batch)(void* h, const int32_t* cols, int32_t ncols, const double* means, int64_t k0, int32_t nlags,
                              double* out);
    int (*kde_lag_sums_batch)(void* h, const int32_t* cols, int32_t ncols, const double* inv4s2, const int64_t* lags,
                              int32_t nlags, double* out);
    int (*kde_lag_sums)(void* h, int32_t col, double inv4s2, const int64_t* lags, int32_t nlags, double* out);
    const char* (*last_error)(void* h);
    // a further context on h's device over the same resident samples (its own stream and scratch) / its end
    int (*create_aux)(void* h, void** aux);
    int (*destroy_aux)(void* aux);
    // gd_kopt2d in two stream-ordered stages (gdhip.h): everything up to the functionals on h's stream, returning at once;
    // get_h for those rows on ANOTHER context's stream behind the ticket, blocking until the rows are on the host
    int (*kopt2d_enqueue)(void* h, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr,
                          const double* fallback_t, const double* corr, void* d_rows, int32_t* ticket);
    int (*kopt2d_finish)(void* h, void* stage_a_h, int32_t ticket, int32_t B, void* d_rows, double* out);
    // the context's communicator (gd_comm_init): number of ranks (0: none) / sum all-reduce of a host vector
    int (*comm_world)(void* h);
    int (*comm_allreduce_sum)(void* h, double* inout, int64_t count);
    // the (idle) context's stream re-created with high (1) / normal (0) / low (-1) priority
    int (*stream_priority)(void* ctx, int level);
    // u16 index columns of several sample columns in ONE launch (gd_prebin_batch; blocking).  May be null: `prebin` per column
    int (*prebin_batch)(void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                        void* const* d_idx);
};

// ---- numpy / CPython scalar semantics --------------------------------------------------------------------------------
// `x ** y` on Python floats and numpy float64 scalars is libm pow(x, y).  The exponent goes through a volatile so that
// no compiler rewrites pow(x, 2.0) as x * x (glibc's pow is not correctly rounded in every case; numpy ARRAY squares,
// written x * x below, are).
static inline double py_pow(double x, double y) {
    volatile double e = y;
    return pow(x, e);
}
static inline double np_minimum(double a, double b) { return (a < b || isnan(a)) ? a : b; }
static inline double np_maximum(double a, double b) { return (a > b || isnan(a)) ? a : b; }
static inline double np_sign(double x) { return isnan(x) ? x : (x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0)); }

// Smallest 2^a * {1,3,5,9,15} (a >= 4) >= n: the frame ladder of the convolution (density2d.hip).
static inline int frame_size(int n) {
    long long best = 1LL << 40;
    const int odd[5] = {1, 3, 5, 9, 15};
    for (int a = 4; a < 28; ++a)
        for (int q = 0; q < 5; ++q) {
            const long long v = (1LL << a) * odd[q];
            if (v >= n && v < best) best = v;
        }
    return (int)best;
}

// np.linalg.cholesky + np.linalg.inv of a 2 x 2 covariance exactly as LAPACK (OpenBLAS potrf / getrf + getrs)
// evaluates them -- reciprocal scaling of the sub-diagonal, partial pivoting in the inverse -- followed by
// mcsamples.py:1352-1356: S *= ichol[0,0]; r = ichol[1,:] / ichol[0,0].  Checked against numpy on random matrices
// (tests/test_native_batch.py).  Returns false when the matrix is not positive definite (numpy raises LinAlgError).
static inline bool chol_shear(double c00, double c10, double c11, double S[4], double r[2]) {
    if (!(c00 > 0)) return false;
    const double l00 = sqrt(c00);
    const double l10 = c10 * (1.0 / l00);
    const double d = c11 - l10 * l10;
    if (!(d > 0)) return false;
    const double l11 = sqrt(d);
    double i00, i10, i11;
    if (fabs(l10) > fabs(l00)) {  // row interchange in dgetrf
        const double l = l00 * (1.0 / l10);
        const double u11 = 0.0 - l * l11;
        i10 = 1.0 * (1.0 / u11);
        i00 = (0.0 - l11 * i10) * (1.0 / l10);
        i11 = (0.0 - l * 1.0) * (1.0 / u11);
    } else {
        const double l = l10 * (1.0 / l00);
        i00 = 1.0 * (1.0 / l00);
        i10 = (0.0 - l * 1.0) * (1.0 / l11);
        i11 = 1.0 * (1.0 / l11);
    }
    S[0] = l00 * i00, S[1] = 0.0 * i00, S[2] = l10 * i00, S[3] = l11 * i00;
    r[0] = i10 / i00, r[1] = i11 / i00;
    return true;
}

// ---- per-pair scalars of get2DDensityGridData (mcsamples.py:1794-1822) ----------------------------------------------
struct PairScalars {
    std::vector<int32_t> jx, jy, F, warn;
    std::vector<int64_t> nbin2D;
    std::vector<double> actual, corr;
};

static inline void pair_scalars(const gd_batch2d_settings& s, int n, const double* corrmat, const int32_t* pairs, int P,
                                PairScalars& ps) {
    ps.jx.resize(P), ps.jy.resize(P), ps.F.resize(P), ps.warn.assign(P, 0), ps.nbin2D.resize(P), ps.actual.resize(P),
        ps.corr.resize(P);
    const int base_F = s.fine_bins_2D;
    for (int k = 0; k < P; ++k) {
        const int a = pairs[2 * k], b = pairs[2 * k + 1];
        ps.jx[k] = a, ps.jy[k] = b;
        const double actual = corrmat[(size_t)b * n + a];  // correlationMatrix[j2, j]
        double c = actual;
        if (fabs(fabs(c) - 1.0) <= 1e-8) {  // "Parameters are 100% correlated" (mcsamples.py:1798-1803)
            ps.warn[k] |= 1;
            c = np_sign(c) * s.max_corr_2D;
        }
        if (fabs(c) < 0.1) c = 0.0;
        const double m = np_minimum(s.max_corr_2D, fabs(c));
        const double angle_scale = np_maximum(0.2, sqrt(1 - m * m));
        ps.nbin2D[k] = (int64_t)nearbyint((double)s.num_bins_2D / angle_scale);
        const int64_t scaled = 192 * (int64_t)(3 / angle_scale) / 3;
        ps.F[k] = (c != 0 && base_F < scaled && (int64_t)(1 / angle_scale) > 1) ? (int32_t)scaled : base_F;
        ps.actual[k] = actual, ps.corr[k] = c;
    }
}

// binmin / binmax of _binSamples (mcsamples.py:1486-1496)
static inline void bin_edges(const gd_param2d& p, double* binmin, double* binmax) {
    const double border = (p.range_max - p.range_min) * 0.1;
    double lo = np_minimum(p.param_min, p.range_min);
    if (!p.has_limits_bot) lo = lo - border;
    double hi = np_maximum(p.param_max, p.range_max);
    if (!p.has_limits_top) hi = hi + border;
    *binmin = lo, *binmax = hi;
}

// ---- effective sample numbers (chains.py:477-574 via mcsamples.py:1230-1235) -------------------------------------------
struct NeffInput {
    int64_t N;
    double norm, sum_w2;
};

// the scalar half of getEffectiveSamplesGaussianKDE given the seven batched lag sums; `lag_sum(k)` fetches another lag
static inline int neff_from_lags(const NeffInput& in, int64_t maxoff, double min_corr, const double* sums, int nseed_tail,
                                 const std::function<int(int64_t, double*)>& lag_sum, double* neff_out) {
    const int64_t N = in.N;
    if (maxoff > N / 10) maxoff = N / 10;
    const int64_t uncorr_len = N / 2;
    int64_t nav = 0;
    for (int64_t k = uncorr_len; k < uncorr_len + 5; ++k) nav += N - k;
    double s5 = 0.0;
    for (int q = 0; q < 5; ++q) s5 += sums[q];
    const double uncorr_term = s5 / (double)nav;
    const double n = (double)N;
    std::map<int64_t, double> cache;
    for (int q = 0; q < nseed_tail; ++q) cache[q + 1] = sums[5 + q];  // lags 1, 2
    int rc = 0;
    auto corr_k = [&](int64_t k) -> double {
        auto it = cache.find(k);
        double v;
        if (it == cache.end()) {
            v = NAN;
            const int e = lag_sum(k, &v);
            if (e) rc = e;
            cache[k] = v;
        } else {
            v = it->second;
        }
        return v - (n - (double)k) * uncorr_term;
    };
    const double corr0 = in.sum_w2;
    const double threshold = min_corr * corr0;
    const double c1 = corr_k(1);
    double Nn;
    if (c1 < threshold) {
        Nn = corr0;
    } else {
        const double c2 = corr_k(2);
        if (c2 > threshold) {
            int64_t max_k = maxoff;
            while (max_k > 10) {
                if (corr_k(max_k / 3) >= threshold) break;
                max_k /= 3;
            }
            const int64_t step_size = max_k < 20 ? 1 : max_k / 10;
            double cum_sum = c1 + c2;
            for (int64_t k = 3; k < maxoff + 1; k += step_size) {
                const double test_val = corr_k(k);
                if (test_val < threshold) break;
                cum_sum += k > 3 ? test_val * (double)step_size : (test_val * (double)step_size) / 2;
            }
            Nn = corr0 + 2 * cum_sum;
        } else {
            Nn = corr0 + 2 * c1;
        }
    }
    if (rc) return rc;
    *neff_out = py_pow(in.norm, 2.0) / Nn;
    return 0;
}

// ---- the bandwidth plan (mcsamples.py:1325-1409) ----------------------------------------------------------------------
struct Plan {
    std::vector<int8_t> branch;        // 0 A, 1 B, 2 C
    std::vector<uint8_t> has_limits;   // parx.has_limits or pary.has_limits
    std::vector<double> rangex, rangey, ratio, neff, fallback_t;
    // branch A only (indexed by pair; valid where branch == 0)
    std::vector<int32_t> si, sj;       // the sheared pair's columns (swapped when y carries the limit)
    std::vector<uint8_t> swap, has_imin, has_imax;
    std::vector<double> imin, imax, S, r;  // S: 4 per pair, r: 2 per pair
};

static inline bool has_limits(const gd_param2d& p) { return p.has_limits_bot || p.has_limits_top; }

static inline int make_plan(const gd_batch2d_settings& s, const gd_param2d* par, int n, const double* cov,
                            const PairScalars& ps, const std::vector<double>& rngx, const std::vector<double>& rngy,
                            double min_corr, Plan& pl, std::string& err) {
    const int P = (int)ps.jx.size();
    pl.branch.resize(P), pl.has_limits.resize(P), pl.rangex = rngx, pl.rangey = rngy, pl.ratio.resize(P);
    pl.neff.assign(P, NAN), pl.fallback_t.assign(P, NAN);
    pl.si.assign(P, -1), pl.sj.assign(P, -1), pl.swap.assign(P, 0), pl.has_imin.assign(P, 0), pl.has_imax.assign(P, 0);
    pl.imin.assign(P, NAN), pl.imax.assign(P, NAN), pl.S.assign((size_t)4 * P, NAN), pl.r.assign((size_t)2 * P, NAN);
    for (int k = 0; k < P; ++k) {
        const gd_param2d &px = par[ps.jx[k]], &py = par[ps.jy[k]];
        const bool limx = has_limits(px), limy = has_limits(py);
        pl.has_limits[k] = limx || limy;
        const bool do_correlated = !limx || !limy;
        const double c = ps.actual[k], absc = fabs(c);
        const bool is_A = (min_corr < absc) && (absc <= s.max_corr_2D) && do_correlated;
        const bool is_B = !is_A && ((absc > s.max_corr_2D) || (!do_correlated && (c > 0.8)));
        pl.branch[k] = is_A ? 0 : (is_B ? 1 : 2);
        pl.ratio[k] = np_minimum(py.sigma_range / rngy[k], px.sigma_range / rngx[k]);
        if (!is_A) continue;
        int i = ps.jx[k], j = ps.jy[k];
        if (px.has_limits_bot) pl.has_imin[k] = 1, pl.imin[k] = px.range_min;
        if (px.has_limits_top) pl.has_imax[k] = 1, pl.imax[k] = px.range_max;
        if (limy) {
            std::swap(i, j);
            pl.swap[k] = 1;
            if (py.has_limits_bot) pl.has_imin[k] = 1, pl.imin[k] = py.range_min;
            if (py.has_limits_top) pl.has_imax[k] = 1, pl.imax[k] = py.range_max;
        }
        pl.si[k] = i, pl.sj[k] = j;
        if (!chol_shear(cov[(size_t)i * n + i], cov[(size_t)j * n + i], cov[(size_t)j * n + j], &pl.S[(size_t)4 * k],
                        &pl.r[(size_t)2 * k])) {
            err = "Matrix is not positive definite";  // numpy.linalg.LinAlgError out of np.linalg.cholesky
            return GD_ERR_BADARG;
        }
    }
    return 0;
}

// N_eff per pair (min of the two parameters', mcsamples.py:1329-1331) and the fallback time of branch C (:1396-1397)
static inline void fill_plan(const gd_param2d* par, const PairScalars& ps, Plan& pl, const double* pair_neff = nullptr) {
    const int P = (int)ps.jx.size();
    for (int k = 0; k < P; ++k) {
        // (pair_neff: the caller's own estimate per pair -- use_effective_samples_2D, mcsamples.py:1322-1328)
        pl.neff[k] = pair_neff ? pair_neff[k] : np_minimum(par[ps.jx[k]].neff, par[ps.jy[k]].neff);
        if (pl.branch[k] == 2) pl.fallback_t[k] = py_pow(pl.ratio[k] / py_pow(pl.neff[k], 1.0 / 6), 2.0);
    }
}

// ---- persistent state of a context's batched calls --------------------------------------------------------------------
struct IdxCol {
    void* ptr = nullptr;
    double binmin = NAN, width = NAN;
    bool valid = false;
};

struct Pending {  // a call whose result copies may still be in flight
    std::vector<void*> blocks;
    int32_t tok_main = -1, tok_twin = -1;
    void* twin = nullptr;
    bool live = false;
};

struct State {
    std::mutex mu;
    std::vector<std::pair<void*, int64_t>> free_blocks;  // (device pointer, capacity)
    std::map<void*, int64_t> capacity;                   // of every block handed out
    int64_t cached_bytes = 0;
    std::map<std::tuple<int, int, int>, IdxCol> idx;     // (column, F, 1 = bytes / 2 = u16)
    Pending prev;                                        // the last lazily delivered call
    void* aux = nullptr;                                 // third context (stream) of large calls: shear chain, get_h
    void* aux2 = nullptr;                                // fourth: the DEFERRED shear chain + up-scaled classes (round 6)
    int64_t exchanges_entered = 0;                       // N_eff collectives the library ENTERED over the communicator (whether or
                                                         // not the call went on to succeed): gd_batch2d_exchanges
    static constexpr int64_t kCacheLimit = 16LL << 30;
};

struct Pool {
    State& st;
    const Ops& ops;
    void* h;
    void* take(int64_t bytes, int* rc) {
        {
            std::lock_guard<std::mutex> g(st.mu);
            int best = -1;
            for (size_t k = 0; k < st.free_blocks.size(); ++k) {
                const int64_t cap = st.free_blocks[k].second;
                if (cap >= bytes && cap <= 2 * bytes + (1 << 20) && (best < 0 || cap < st.free_blocks[best].second)) best = (int)k;
            }
            if (best >= 0) {
                void* p = st.free_blocks[best].first;
                st.cached_bytes -= st.free_blocks[best].second;
                st.free_blocks.erase(st.free_blocks.begin() + best);
                return p;
            }
        }
        void* p = nullptr;
        int e = ops.dev_alloc(h, bytes, &p);
        if (e == GD_ERR_NOMEM) {  // hand the cached blocks back and retry once
            drop_cached();
            e = ops.dev_alloc(h, bytes, &p);
        }
        if (e) {
            *rc = e;
            return nullptr;
        }
        std::lock_guard<std::mutex> g(st.mu);
        st.capacity[p] = bytes;
        return p;
    }
    void give(void* p) {
        if (!p) return;
        std::unique_lock<std::mutex> g(st.mu);
        const int64_t cap = st.capacity[p];
        if (st.cached_bytes + cap > State::kCacheLimit) {
            st.capacity.erase(p);
            g.unlock();
            ops.dev_free(h, p);
            return;
        }
        st.free_blocks.emplace_back(p, cap);
        st.cached_bytes += cap;
    }
    void drop_cached() {
        std::vector<std::pair<void*, int64_t>> blocks;
        {
            std::lock_guard<std::mutex> g(st.mu);
            blocks.swap(st.free_blocks);
            st.cached_bytes = 0;
            for (auto& b : blocks) st.capacity.erase(b.first);
        }
        for (auto& b : blocks) ops.dev_free(h, b.first);
    }
};

static inline int complete_pending(State& st, const Ops& ops, void* h, Pending& pd) {
    if (!pd.live) return 0;
    int rc = 0;
    if (pd.tok_main >= 0) rc = ops.copy_wait(h, pd.tok_main);
    if (!rc && pd.twin && pd.tok_twin >= 0) rc = ops.copy_wait(pd.twin, pd.tok_twin);
    Pool pool{st, ops, h};
    for (void* p : pd.blocks) pool.give(p);
    pd.blocks.clear();
    pd.live = false;
    return rc;
}

static inline int finish_all(State& st, const Ops& ops, void* h) { return complete_pending(st, ops, h, st.prev); }

static inline void invalidate_index_columns(State& st) {
    std::lock_guard<std::mutex> g(st.mu);
    for (auto& kv : st.idx) kv.second.valid = false;
}

// frees every device block the state holds (context teardown / new sample set)
static inline void release_all(State& st, const Ops& ops, void* h) {
    finish_all(st, ops, h);
    Pool pool{st, ops, h};
    {
        std::lock_guard<std::mutex> g(st.mu);
        for (auto& kv : st.idx)
            if (kv.second.ptr) st.free_blocks.emplace_back(kv.second.ptr, st.capacity[kv.second.ptr]);
        st.idx.clear();
    }
    pool.drop_cached();
    if (st.aux) {  // (attached to the sample set that is going away)
        ops.destroy_aux(st.aux);
        st.aux = nullptr;
    }
    if (st.aux2) {
        ops.destroy_aux(st.aux2);
        st.aux2 = nullptr;
    }
}

// ---- the call ----------------------------------------------------------------------------------------------------------
struct Call {
    State& st;
    const Ops& ops;
    void* h;     // main context: first stream
    void* twin;  // second stream (may be null)
    const gd_batch2d_settings& s;
    gd_param2d* par;
    int n;
    const double* corrmat;
    const double* cov;
    const double* lag_probe;
    const int32_t* pairs;
    int P;
    gd_neff_exchange_fn exchange;
    void* exchange_user;
    double* grids;
    int64_t grids_doubles;
    int32_t* status_pinned;
    double* meta;
    double* levels;
    int32_t* level_status;
    std::string err;

    Call(State& st_, const Ops& ops_, void* h_, void* twin_, const gd_batch2d_settings& s_, gd_param2d* par_, int n_,
         const double* corr_, const double* cov_, const double* lag_probe_, const int32_t* pairs_, int P_,
         gd_neff_exchange_fn ex_, void* ex_user_, double* grids_, int64_t grids_doubles_, int32_t* status_, double* meta_,
         double* levels_, int32_t* level_status_)
        : st(st_), ops(ops_), h(h_), twin(twin_), s(s_), par(par_), n(n_), corrmat(corr_), cov(cov_), lag_probe(lag_probe_),
          pairs(pairs_), P(P_), exchange(ex_), exchange_user(ex_user_), grids(grids_), grids_doubles(grids_doubles_),
          status_pinned(status_), meta(meta_), levels(levels_), level_status(level_status_), pool{st_, ops_, h_} {}

    int64_t N = 0;
    bool unit_weights = true;
    Pool pool;
    PairScalars ps;
    std::vector<int> used;
    std::vector<double> bmin, bmax;  // per column
    std::vector<double> fwx, fwy;    // per pair
    std::vector<int> F_list;         // grid sizes in order of first appearance
    std::map<int, std::vector<int>> classes;  // F -> pair indices, ascending
    std::map<int, void*> hists;               // F -> device histograms of the class, in member order
    Plan plan;
    struct Shear {
        void* d_rot = nullptr;
        std::vector<int> A;
        std::vector<double> r1s, r2s;
    } shear;
    bool have_shear = false;
    bool byte_index_weights = false;  // real weights: the byte-index route is served too (round 6: gd_hist2d_prebinned8's sorted form)
    // convolution state
    std::vector<double> rx, ry, cc, smooth, W;  // W: P x 3
    std::vector<int64_t> winw;
    std::vector<int32_t> flags, group;
    std::vector<void*> call_blocks;  // everything to release once the copies have landed
    std::vector<void*> bin_stale;    // a main-class buffer whose second launch failed (under enq_mu; joins call_blocks)
    std::mutex enq_mu;
    // three-stream choreography: the N_eff launch (2000 resident blocks for ~3 ms: whatever is enqueued behind it on another
    // stream waits for a wave slot) holds back until the binning thread is about to enqueue its chain -- the binning is on
    // the critical path, the N_eff values are needed after it (round 6: a race the binning used to win by ~0.2 ms)
    std::atomic<int> bin_launching{0};
    bool hold_neff_for_binning = false;
    // the sheared pairs (and the up-scaled grid classes behind them) are convolved LAST: their O(N) chain -- as long as the
    // main class's -- may wait until the main binning has run instead of sharing the machine with it (the main class's
    // histograms gate the optimiser and with it the first result copies: single-triangle latency)
    std::atomic<int> bin_done{0};
    bool shear_after_binning = false;
    // Deferred shear chain (round 6, single-triangle latency): the first result copies can leave once the FIRST optimiser
    // part's pairs are convolved, and 642 MB of grids then take ~13 ms of PCIe -- so what that first part has to wait for
    // decides when a triangle is delivered.  The sheared pairs (branch A: min/max of the sheared coordinate, a re-binning pass
    // over the fp64 samples) and the up-scaled grid classes used to be joined BEFORE any optimiser launch and the sheared pairs
    // rode in the first part; now their chain runs on a context of its own, starts when the main class has been binned,
    // the sheared pairs ride in the LAST base part, and the main thread joins the chain only when it builds the first launch
    // that needs it.  Results are unchanged bit for bit (a pair's grid does not depend on the part it is computed in).
    std::future<int> shear_future;
    bool shear_deferred = false;
    std::atomic<bool> shear_joined{true};
    std::mutex shear_mu;  // (join_shear is entered by the staging thread and, for the last report, by the enqueuing thread)
    // The chain's two halves are awaited separately: a launch that carries sheared rows needs the SHEARED HISTOGRAMS only
    // (shear_hist_state: 0 running, 1 done, 2 failed with shear_hist_rc), not the up-scaled grid classes the chain bins behind
    // them -- the last base part's stage A used to start 2.5 ms later than its rows existed (stream timeline of a C3 step:
    // sheared histograms done at 16.9 ms, the up-scaled classes at 19.4, the last part's first kernel at 19.6).
    std::atomic<int> shear_hist_state{1};
    int shear_hist_rc = 0;
    // ... and it starts only when the FIRST optimiser part's convolution has been enqueued (its stage A, get_h and first
    // batches get the machine ahead of the chain's 1024-thread blocks; the chain then runs beside the later parts): same-box
    // A/B, delivered triangle: chain at once 29.4-29.9 ms, released on the first part's bandwidths 28.8-29.5, on its
    // convolution 28.1-28.9; the stream of triangles 21.7 -> 21.9.  Released at once when the first launch itself needs the
    // chain, when nothing is staged, and on every error path (and after 20 ms whatever happens).
    std::atomic<int> first_part_done{0};
    // ... and, since the end of round 6, only when that convolution has RUN: the chain's 1024-thread blocks take every CU for
    // ~3 ms, and started on the enqueue they ran beside the first part's convolution -- the triangle's first grids, which the
    // 13 ms of result copies wait for.  Found through a profiler artefact: under rocprofv3's kernel trace the enqueue of the
    // first batches takes 2.5 ms instead of 0.3, the chain started that much later, and a delivered triangle took 24.6 ms instead
    // of 26.5 (the stream of triangles 22.4 instead of 19.9).  A timed hold showed the same (GDHIP_BATCH_SHEAR_HOLD_US = 1000 /
    // 2000 / 3000 / 4500: delivered 26.1 / 24.6 / 24.5 / 24.3 ms, stream 20.4 / 20.6 / 21.6 / 22.2 against 26.5 / 19.9 without).
    // The release is the status word of the first report's last batch: enqueue_batch pre-sets a batch's words to a sentinel, the
    // batch's last launch (a copy kernel into page-locked memory) overwrites them, and the chain's thread polls the word.
    // GDHIP_BATCH_SHEAR_ON_ENQUEUE=1: the release of the enqueue, as before -- which is also what a STREAM of calls gets
    // (settings.results_in_flight: this call's first grids queue behind the previous call's copies, its delivery does not wait
    // for them, and the chain beside the first convolution is the better use of the machine: 19.9 against 21.0 ms per triangle).
    static constexpr int32_t kStatusPending = 0x7fffffff;
    std::atomic<const volatile int32_t*> first_part_flag{nullptr};
    int reports_enqueued = 0;  // (enqueuing thread only)
    // Main binning in two launches (round 6, single-triangle latency): the members of the base grid's class are put in the
    // order the optimiser takes them (its own pairs first, the sheared and rule-of-thumb pairs behind), the rows of the FIRST
    // optimiser part are binned by the fused prebin + histogram call, the other rows by a second histogram launch behind it
    // -- and stage A of the first part starts when the first launch has run instead of waiting for both.  Everything else
    // (the later parts, every convolution, the deferred chain) joins the whole binning as before.  A pair's histogram does
    // not depend on its row or on the launch that fills it: results unchanged bit for bit.
    std::shared_future<int> bin_future;
    std::mutex bin_mu;
    std::atomic<int> bins_first_done{0};
    int bins_first_rows = 0;  // rows [0, bins_first_rows) of the main class's buffer are the first launch's; 0: one launch
    int bin_rc = 0;
    bool bin_joined = true;
    int join_binning() {
        std::lock_guard<std::mutex> g(bin_mu);
        if (!bin_joined) {
            bin_joined = true;
            bin_rc = bin_future.valid() ? bin_future.get() : 0;
            mark("binning: joined");
        }
        return bin_rc;
    }
    // the size of a part of the base grid's optimiser launches (total = its own pairs + the sheared pairs)
    size_t base_part_size(size_t total, int F) const {
        // Two parts: the second's DCT / fixed point beside the first's get_h and convolution.  More, smaller
        // parts were measured (3 / 4 / 12: 31.2 / 32.1 / 31.4 ms per C3 step against 30.5 with two): the chip is
        // busy either way, and a part below two blocks per CU leaves the fixed-point kernel's tail exposed.
        size_t want = 2;  // GDHIP_KOPT_PARTS: tuning knob
        if (const char* e = getenv("GDHIP_KOPT_PARTS")) want = (size_t)std::max(1, atoi(e));
        size_t part = std::max<size_t>((size_t)s_kopt_split_min(), (total + want - 1) / want);
        // the fixed-point kernel holds one pair per CU (its matrix lives in the CU's registers and LDS): parts of
        // a whole number of rounds over the 256 CUs leave no extra, mostly empty round (1279 pairs as 640 + 639
        // are 3 + 3 rounds, as 512 + 767 they are 2 + 3)
        size_t align = 256;  // GDHIP_KOPT_PART_ALIGN: tuning knob (0: off)
        if (const char* e = getenv("GDHIP_KOPT_PART_ALIGN")) align = (size_t)std::max(0, atoi(e));
        if (align > 1 && F == 256 && part > align) part = std::max<size_t>(align, (part + align / 2 - 1) / align * align);
        return part;
    }
    // the FIRST part may be smaller than the others (GDHIP_KOPT_FIRST_PART: tuning knob; 0 = like the others)
    static size_t first_part_size(size_t part) {
        if (const char* e = getenv("GDHIP_KOPT_FIRST_PART")) {
            const size_t f = (size_t)std::max(0, atoi(e));
            if (f > 0 && f < part) return f;
        }
        return part;
    }
    static int shear_hold_us() {
        const char* e = getenv("GDHIP_BATCH_SHEAR_HOLD_US");
        return e ? atoi(e) : 0;
    }
    int join_shear(bool release = false) {
        // release: the caller is not going to enqueue a first part the chain could wait for (error paths, a call whose only
        // launch needs the chain).  The staging thread's ordinary join does NOT release: by then the first part is enqueued
        // and the finisher releases the chain when that part's bandwidths are final.
        if (release) first_part_done.store(1);
        std::lock_guard<std::mutex> g(shear_mu);
        if (shear_joined.load()) return 0;
        const int e = shear_future.valid() ? shear_future.get() : 0;
        shear_joined.store(true);
        mark("shear: joined");
        return e;
    }
    int64_t grid_off = 0;
    int status_at = 0;
    int batch_no = 0;
    std::vector<void*> conv_ctxs;
    std::vector<int> side_classes;
    std::vector<int> order;  // grid sizes, largest class (in bytes) first

    // host-side timeline of the call (GDHIP_BATCH_LOG=1 prints it to stderr when the call returns)
    std::mutex tl_mu;
    std::vector<std::pair<double, std::string>> tl;
    bool tl_on = getenv("GDHIP_BATCH_LOG") != nullptr;
    static double now_ms() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    }
    void mark(const char* what, int a = -1, int b = -1) {
        if (!tl_on) return;
        char buf[160];
        if (a >= 0 && b >= 0)
            snprintf(buf, sizeof buf, "%s (%d, %d)", what, a, b);
        else if (a >= 0)
            snprintf(buf, sizeof buf, "%s (%d)", what, a);
        else
            snprintf(buf, sizeof buf, "%s", what);
        std::lock_guard<std::mutex> g(tl_mu);
        tl.emplace_back(now_ms(), buf);
    }
    void dump_timeline() {
        if (!tl_on || tl.empty()) return;
        std::sort(tl.begin(), tl.end());
        fprintf(stderr, "---- gd_density2d_batch host timeline (ms)\n");
        for (auto& e : tl) fprintf(stderr, "%9.3f  %s\n", e.first - tl[0].first, e.second.c_str());
    }

    std::mutex err_mu;
    int fail(int code, const std::string& msg) {
        std::lock_guard<std::mutex> g(err_mu);
        if (err.empty()) err = msg;
        return code;
    }
    int dev_fail(int code, void* ctx) {
        const char* m = ops.last_error(ctx);
        return fail(code, m && *m ? m : "device call failed");
    }
#define GDB_DEV(ctx_, call_)                       \
    do {                                           \
        const int rc__ = (call_);                  \
        if (rc__) return dev_fail(rc__, (ctx_));   \
    } while (0)
#define GDB_TRY(call_)            \
    do {                          \
        const int rc__ = (call_); \
        if (rc__) return rc__;    \
    } while (0)

    double* M(int k) { return meta + (size_t)k * GD_BATCH2D_META; }

    // -- N_eff of the listed columns that have none yet (the batched form of _get1DNeff, mcsamples.py:1230-1235)
    int neff_batch(const std::vector<int>& js, bool owned_only, double min_corr = 0.05) {
        std::vector<int32_t> todo;
        for (int j : js)
            if (isnan(par[j].neff) && (!owned_only || (par[j].owned & 1))) todo.push_back(j);
        if (todo.empty()) return 0;
        mark("neff: start", (int)todo.size());
        if (s.uncorrelated_sampler) {
            for (int j : todo) par[j].neff = py_pow(s.norm, 2.0) / s.sum_w2;
            return 0;
        }
        const int m = (int)todo.size();
        const int64_t max_off = N / 10;
        const int nl = (int)std::min<int64_t>(8, max_off + 1);
        std::vector<double> lag0((size_t)m * nl);
        bool have = lag_probe != nullptr && nl == 8;
        if (have)
            for (int row = 0; row < m && have; ++row)
                for (int k = 0; k < nl; ++k) {
                    lag0[(size_t)row * nl + k] = lag_probe[(size_t)todo[row] * 8 + k];
                    if (isnan(lag0[(size_t)row * nl + k])) have = false;
                }
        if (!have) {
            std::vector<double> means(m);
            for (int row = 0; row < m; ++row) means[row] = par[todo[row]].mean;
            GDB_DEV(h, ops.autocov_lags_batch(h, todo.data(), m, means.data(), 0, nl, lag0.data()));
        }
        std::vector<double> kstd(m), inv4s2(m);
        std::vector<int64_t> maxoffs(m);
        bool all_uncorrelated_at_lag_1 = true;  // by the probe: then the speculative corr_k(2) is left out of the launch
        for (int row = 0; row < m; ++row) {
            const gd_param2d& p = par[todo[row]];
            double c[8];
            for (int k = 0; k < nl; ++k) c[k] = lag0[(size_t)row * nl + k] / (double)(N - k) / p.var;
            int first_below = -1;
            for (int k = 0; k < nl; ++k)
                if (!(c[k] > min_corr * c[0])) {
                    first_below = k;
                    break;
                }
            if (first_below != 1) all_uncorrelated_at_lag_1 = false;
            double corrlen;
            if (first_below >= 0) {
                double sum = 0.0;
                for (int k = 1; k < first_below; ++k) sum += c[k];
                corrlen = c[0] + 2 * sum;
            } else if (nl == max_off + 1) {
                corrlen = c[0];
            } else {
                char buf[160];
                snprintf(buf, sizeof buf, "column %d: the chain's correlation outlasts the 8-lag probe (getCorrelationLength's long route)",
                         (int)todo[row]);
                return fail(GD_BATCH2D_NEED_NEFF, buf);
            }
            const double sr = p.sigma_range;
            kstd[row] = ((isnan(sr) || sr == 0.0) ? p.err : sr) * 0.2;  // (par.sigma_range or self.sddev[j]) * 0.2
            inv4s2[row] = 1.0 / (4 * py_pow(kstd[row], 2.0));
            maxoffs[row] = (int64_t)(corrlen * 1.5) + 4;
        }
        std::vector<int64_t> lags;
        const int64_t uncorr_len = N / 2;
        for (int64_t k = uncorr_len; k < uncorr_len + 5; ++k) lags.push_back(k);
        // corr_k(1) always; corr_k(2) -- needed only by a chain still correlated at lag 1 (chains.py:541-545) -- rides along
        // unless the probe shows every column below the threshold at lag 1 already: six exponentials per sample instead of
        // seven (the launch is bound by them); a column the probe misjudged fetches its lag 2 by itself (neff_from_lags)
        int ntail = 0;
        for (int64_t k : {1, 2})
            if (k <= N / 10 && (k == 1 || !all_uncorrelated_at_lag_1)) lags.push_back(k), ++ntail;
        const int L = (int)lags.size();
        std::vector<double> sums((size_t)m * L);
        mark("neff: probe done");
        if (hold_neff_for_binning) {
            const auto t0 = std::chrono::steady_clock::now();
            while (!bin_launching.load() && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2)) std::this_thread::yield();
            int hold_us = 120;  // (its launches go out)  GDHIP_BATCH_NEFF_HOLD_US: tuning knob
            if (const char* e = getenv("GDHIP_BATCH_NEFF_HOLD_US")) hold_us = atoi(e);
            if (bin_launching.load()) std::this_thread::sleep_for(std::chrono::microseconds(hold_us));
            mark("neff: binning chain enqueued");
        }
        GDB_DEV(h, ops.kde_lag_sums_batch(h, todo.data(), m, inv4s2.data(), lags.data(), L, sums.data()));
        mark("neff: lag sums done");
        const NeffInput in{N, s.norm, s.sum_w2};
        for (int row = 0; row < m; ++row) {
            const int col = todo[row];
            const double i4 = inv4s2[row];
            auto lag_sum = [&](int64_t k, double* out) -> int { return ops.kde_lag_sums(h, col, i4, &k, 1, out); };
            double v = NAN;
            const int e = neff_from_lags(in, maxoffs[row], min_corr, &sums[(size_t)row * L], ntail, lag_sum, &v);
            if (e) return dev_fail(e, h);
            par[col].neff = v;
        }
        return 0;
    }

    // -- multi-rank: the other ranks' N_eff values, then whatever nobody owned (mcsamples._neff_complete)
    int neff_exchange(bool* exchanged) {
        if (exchange && !*exchanged) {
            *exchanged = true;
            std::vector<double> v(n);
            for (int j = 0; j < n; ++j) v[j] = par[j].neff;
            if (exchange(exchange_user, v.data(), n)) return fail(GD_ERR_HIP, "N_eff exchange failed");
            for (int j = 0; j < n; ++j)
                if (isnan(par[j].neff)) par[j].neff = v[j];
        } else if (!exchange && s.comm_exchange && !*exchanged) {
            // the library's own exchange: every parameter is owned by exactly one rank, so a sum all-reduce of the owners'
            // values (zero elsewhere) delivers all of them; one collective per call on every rank, whatever its share needs
            *exchanged = true;
            if (!ops.comm_world || ops.comm_world(h) < 1) return fail(GD_ERR_BADARG, "comm_exchange without a communicator (gd_comm_init)");
            {  // counted BEFORE the collective is issued: a caller whose call fails later must know that this rank took part
                std::lock_guard<std::mutex> g(st.mu);
                ++st.exchanges_entered;
            }
            std::vector<double> v(n, 0.0);
            for (int j = 0; j < n; ++j)
                if ((par[j].owned & 1) && !isnan(par[j].neff)) v[j] = par[j].neff;
            GDB_DEV(h, ops.comm_allreduce_sum(h, v.data(), n));
            for (int j = 0; j < n; ++j)
                if (isnan(par[j].neff) && v[j] > 0) par[j].neff = v[j];
        }
        return 0;
    }
    int neff_complete(bool* exchanged) {
        const int rc = neff_exchange(exchanged);
        if (rc || s.pair_neff) return rc;  // (per-pair values from the caller: no parameter's own N_eff is needed)
        return neff_batch(used, false);
    }

    int index_column16(void* ctx, int j, int F, void** out) {
        const double fw = (bmax[j] - bmin[j]) / (F - 1);
        IdxCol c;
        {
            std::lock_guard<std::mutex> g(st.mu);
            c = st.idx[std::make_tuple(j, F, 2)];
        }
        if (!(c.valid && c.binmin == bmin[j] && c.width == fw)) {
            if (!c.ptr) {
                int rc = 0;
                c.ptr = pool.take(N * 2 + 64, &rc);
                if (!c.ptr) return dev_fail(rc, h);
            }
            GDB_DEV(ctx, ops.prebin(ctx, j, bmin[j], fw, F, c.ptr));
            c.binmin = bmin[j], c.width = fw, c.valid = true;
            std::lock_guard<std::mutex> g(st.mu);
            st.idx[std::make_tuple(j, F, 2)] = c;
        }
        *out = c.ptr;
        return 0;
    }

    // the stale u16 index columns of a grid-size class in ONE launch (round 6: the up-scaled classes used to cost a launch
    // per column -- 12 launches of 20-70 us between the shear chain and their histograms)
    int index_columns16(void* ctx, const std::vector<int>& cols, int F) {
        if (!ops.prebin_batch) return 0;  // (index_column16 makes them one by one)
        std::vector<int32_t> todo;
        std::vector<double> b0, w;
        std::vector<void*> bufs;
        for (int j : cols) {
            const double fw = (bmax[j] - bmin[j]) / (F - 1);
            IdxCol c;
            {
                std::lock_guard<std::mutex> g(st.mu);
                c = st.idx[std::make_tuple(j, F, 2)];
            }
            if (c.valid && c.binmin == bmin[j] && c.width == fw) continue;
            if (!c.ptr) {
                int rc = 0;
                c.ptr = pool.take(N * 2 + 64, &rc);
                if (!c.ptr) return dev_fail(rc, h);
                std::lock_guard<std::mutex> g(st.mu);
                st.idx[std::make_tuple(j, F, 2)] = c;  // (the block is the column's from now on, valid or not)
            }
            todo.push_back(j), b0.push_back(bmin[j]), w.push_back(fw), bufs.push_back(c.ptr);
        }
        if (todo.size() < 2) return 0;  // nothing, or one column: the single-column entry serves it
        GDB_DEV(ctx, ops.prebin_batch(ctx, todo.data(), (int)todo.size(), b0.data(), w.data(), F, bufs.data()));
        std::lock_guard<std::mutex> g(st.mu);
        for (size_t q = 0; q < todo.size(); ++q) {
            IdxCol& c = st.idx[std::make_tuple((int)todo[q], F, 2)];
            c.binmin = b0[q], c.width = w[q], c.valid = true;
        }
        return 0;
    }

    // -- prebin + batched 2D histograms of every grid-size class on context `ctx` (mcsamples.py:1486-1498, 1724-1728)
    // part 0: every grid-size class; 1: the class with the most pairs only; 2: the others.  Large calls bin the main class
    // on the second stream and the few pairs of the up-scaled classes behind the shear chain on the third: they used to
    // follow the main class on the second stream -- 1.4 ms of small launches between the end of the O(N) phase and the
    // first optimiser kernel, with the other two streams idle.
    int main_class() const {
        int best = F_list.empty() ? 0 : F_list[0];
        for (int F : F_list)
            if (classes.at(F).size() > classes.at(best).size()) best = F;
        return best;
    }
    int binning(void* ctx, int part = 0) {
        const int main_F = main_class();
        for (int F : F_list) {
            if ((part == 1 && F != main_F) || (part == 2 && F == main_F)) continue;
            const std::vector<int>& members = classes.at(F);
            const int B = (int)members.size();
            int rc = 0;
            if (F == 256 && (unit_weights || byte_index_weights) && B >= 64) {
                // the base grid of a unit-weight triangle: byte indices, packed 16-bit counters, one block per pair
                std::vector<int> cols;
                std::vector<char> seen(n, 0);
                for (int k : members)
                    if (!seen[ps.jx[k]]) seen[ps.jx[k]] = 1, cols.push_back(ps.jx[k]);
                for (int k : members)
                    if (!seen[ps.jy[k]]) seen[ps.jy[k]] = 1, cols.push_back(ps.jy[k]);
                // the stale byte index columns and the histograms over them go out back to back: one wait for both
                std::vector<int32_t> todo;
                std::vector<double> b0, w;
                std::vector<void*> bufs;
                {
                    std::lock_guard<std::mutex> g(st.mu);
                    for (int j : cols) {
                        const double fw = (bmax[j] - bmin[j]) / 255;
                        IdxCol& c = st.idx[std::make_tuple(j, 256, 1)];
                        if (c.valid && c.binmin == bmin[j] && c.width == fw) continue;
                        todo.push_back(j), b0.push_back(bmin[j]), w.push_back(fw), bufs.push_back(c.ptr);
                    }
                }
                for (size_t q = 0; q < todo.size(); ++q)
                    if (!bufs[q]) {
                        bufs[q] = pool.take(N + 64, &rc);
                        if (!bufs[q]) return dev_fail(rc, h);
                        std::lock_guard<std::mutex> g(st.mu);
                        IdxCol& c = st.idx[std::make_tuple((int)todo[q], 256, 1)];
                        c.ptr = bufs[q], c.valid = false;
                    }
                std::vector<const void*> ix(B), iy(B);
                {
                    std::lock_guard<std::mutex> g(st.mu);
                    for (int q = 0; q < B; ++q) {
                        ix[q] = st.idx[std::make_tuple((int)ps.jx[members[q]], 256, 1)].ptr;
                        iy[q] = st.idx[std::make_tuple((int)ps.jy[members[q]], 256, 1)].ptr;
                    }
                }
                void* d = pool.take((int64_t)B * 65536 * 8, &rc);
                if (!d) return dev_fail(rc, h);
                std::vector<int64_t> bad(todo.size() + 1, 0);
                mark("binning: prebin8 + hist2d launch", (int)todo.size(), B);
                bin_launching.store(1);
                // (two launches: the first optimiser part's rows, then the others -- see bin_future)
                const int B1 = (F == main_F && bins_first_rows > 0 && bins_first_rows < B) ? bins_first_rows : B;
                int e = ops.prebin8_hist2d(ctx, todo.data(), (int)todo.size(), b0.data(), w.data(), bufs.data(), bad.data(), B1,
                                           ix.data(), iy.data(), d);
                mark("binning: prebin8 + hist2d done", B1);
                if (e == 0 || e == GD_ERR_SOLVER) {
                    std::lock_guard<std::mutex> g(st.mu);
                    for (size_t q = 0; q < todo.size(); ++q) {
                        IdxCol& c = st.idx[std::make_tuple((int)todo[q], 256, 1)];
                        c.binmin = b0[q], c.width = w[q], c.valid = bad[q] == 0;
                    }
                }
                bool published = false;
                if (e == 0 && B1 < B) {
                    {
                        std::lock_guard<std::mutex> g(enq_mu);
                        hists[F] = d;
                    }
                    published = true;
                    bins_first_done.store(1);
                    e = ops.hist2d_prebinned8(ctx, B - B1, ix.data() + B1, iy.data() + B1, (char*)d + (int64_t)B1 * 65536 * 8);
                    mark("binning: the other rows done", B - B1);
                    if (e == 0) continue;
                }
                if (e == 0) {
                    std::lock_guard<std::mutex> g(enq_mu);
                    hists[F] = d;
                    continue;
                }
                if (published) {  // the first part may be reading its rows: the block lives until the call ends
                    std::lock_guard<std::mutex> g(enq_mu);
                    hists.erase(F);
                    bin_stale.push_back(d);
                } else {
                    pool.give(d);
                }
                if (e != GD_ERR_SOLVER) return dev_fail(e, ctx);  // (else: the u16 / u32 path below redoes the class)
            }
            bin_launching.store(1);  // (this class enqueues as it goes)
            {
                std::vector<int> cols;
                std::vector<char> seen(n, 0);
                for (int k : members) {
                    if (!seen[ps.jx[k]]) seen[ps.jx[k]] = 1, cols.push_back(ps.jx[k]);
                    if (!seen[ps.jy[k]]) seen[ps.jy[k]] = 1, cols.push_back(ps.jy[k]);
                }
                GDB_TRY(index_columns16(ctx, cols, F));
            }
            std::vector<const void*> ix(B), iy(B);
            for (int q = 0; q < B; ++q) {
                void* p;
                GDB_TRY(index_column16(ctx, ps.jx[members[q]], F, &p));
                ix[q] = p;
            }
            for (int q = 0; q < B; ++q) {
                void* p;
                GDB_TRY(index_column16(ctx, ps.jy[members[q]], F, &p));
                iy[q] = p;
            }
            void* d = pool.take((int64_t)B * F * F * 8, &rc);
            if (!d) return dev_fail(rc, h);
            const int e = ops.hist2d_prebinned(ctx, B, ix.data(), iy.data(), F, d);
            mark("binning: hist2d_prebinned done", B, F);
            if (e) {
                pool.give(d);
                return dev_fail(e, ctx);
            }
            std::lock_guard<std::mutex> g(enq_mu);
            hists[F] = d;
        }
        bin_launching.store(1);
        return 0;
    }
    // (the thread that runs binning(ctx, 1) sets bins_first_done when binning returns, whatever the route taken)

    // -- branch A of getAutoBandwidth2D (mcsamples.py:1347-1378): min/max of the sheared coordinate and the re-binned
    //    base grid of every sheared pair, two batched launches
    int shear_histograms(void* ctx) {
        const int base_F = s.fine_bins_2D;
        shear.A.clear();
        for (int k = 0; k < P; ++k)
            if (plan.branch[k] == 0) shear.A.push_back(k);
        have_shear = true;
        const int nA = (int)shear.A.size();
        if (!nA) return 0;
        std::vector<int32_t> ci(nA), cj(nA);
        std::vector<double> r0(nA), r1(nA), mm((size_t)2 * nA);
        for (int row = 0; row < nA; ++row) {
            const int k = shear.A[row];
            ci[row] = plan.si[k], cj[row] = plan.sj[k], r0[row] = plan.r[(size_t)2 * k], r1[row] = plan.r[(size_t)2 * k + 1];
        }
        mark("shear: start", nA);
        GDB_DEV(ctx, ops.minmax_affine(ctx, nA, ci.data(), cj.data(), r0.data(), r1.data(), mm.data()));
        mark("shear: minmax done");
        std::vector<double> xmin(nA), dx(nA), ymin(nA), dy(nA);
        shear.r1s.resize(nA), shear.r2s.resize(nA);
        for (int row = 0; row < nA; ++row) {
            const int k = shear.A[row];
            // kde.bin_samples(p1, nbins, range_min=imin, range_max=imax) (kde_bandwidth.py:76-87)
            const double mn = par[plan.si[k]].param_min, mx = par[plan.si[k]].param_max;
            const double delta = mx - mn;
            const double rmin = plan.has_imin[k] ? plan.imin[k] : mn - delta * 0.1;
            const double rmax = plan.has_imax[k] ? plan.imax[k] : mx + delta * 0.1;
            const double R1 = rmax - rmin;
            const double mn2 = mm[(size_t)2 * row], mx2 = mm[(size_t)2 * row + 1];
            const double delta2 = mx2 - mn2;
            const double rmin2 = mn2 - delta2 * 0.1;
            const double R2 = (mx2 + delta2 * 0.1) - rmin2;
            xmin[row] = rmin, dx[row] = R1 / (base_F - 1), ymin[row] = rmin2, dy[row] = R2 / (base_F - 1);
            shear.r1s[row] = R1, shear.r2s[row] = R2;
        }
        int rc = 0;
        shear.d_rot = pool.take((int64_t)nA * base_F * base_F * 8, &rc);
        if (!shear.d_rot) return dev_fail(rc, h);
        GDB_DEV(ctx, ops.hist2d_sheared(ctx, nA, ci.data(), cj.data(), r0.data(), r1.data(), xmin.data(), dx.data(), ymin.data(),
                                        dy.data(), base_F, shear.d_rot));
        mark("shear: histograms done");
        return 0;
    }

    void set_scales(const std::vector<int>& ks) {
        const double a = fabs(s.smooth_scale_2D);
        for (int k : ks) {
            rx[k] = W[(size_t)3 * k] * a / fwx[k];
            ry[k] = W[(size_t)3 * k + 1] * a / fwy[k];
            cc[k] = W[(size_t)3 * k + 2];
            finish_scale(k);
        }
    }
    void finish_scale(int k) {
        smooth[k] = np_maximum(rx[k], ry[k]);
        const double w = nearbyint(2.5 * smooth[k]);  // max(1, int(round(2.5 * smooth_scale)))
        winw[k] = (isfinite(w) && w > 1) ? (int64_t)w : 1;
        if (smooth[k] < 2) ps.warn[k] |= 2;  // "fine_bins_2D not large enough for optimal density"
    }

    // -- convolution of the pairs of one grid-size class (`only`: mask over the pairs, empty = all) on `force_ctx`, or
    //    on the stream the two-stream rules pick; `limit_batches` > 0 stops after that many batches and returns the
    //    remaining ones in `rest` (the caller interleaves classes)
    struct Batch {
        std::vector<int> pos, ks;
    };
    int class_batches(int F, const std::vector<char>& only, std::vector<Batch>& batches) {
        const std::vector<int>& members = classes.at(F);
        std::vector<int> pos_all;
        for (int q = 0; q < (int)members.size(); ++q)
            if (only.empty() || only[members[q]]) pos_all.push_back(q);
        batches.clear();
        if (pos_all.empty()) return 0;
        int64_t mb = (int64_t)(s_max_batch_bytes() / ((double)F * F * 8 * 30));
        int max_batch = (int)std::max<int64_t>(1, std::min<int64_t>(mb, s_max_batch()));
        const int first_batch = s_first_batch();
        // groups in order of first appearance; within a group sub-batches of equal frame size S >= F + 2 winw, ascending
        std::vector<int> gorder;
        for (int q : pos_all) {
            const int g = group[members[q]];
            if (std::find(gorder.begin(), gorder.end(), g) == gorder.end()) gorder.push_back(g);
        }
        for (int g : gorder) {
            std::map<int, std::vector<int>> byS;
            for (int q : pos_all) {
                const int k = members[q];
                if (group[k] != g) continue;
                const int64_t w = winw[k];
                if (w < 1 || w > 2 * (int64_t)F) return fail(GD_ERR_BADARG, "bad window half-width");
                byS[frame_size(F + 2 * (int)w)].push_back(q);
            }
            for (auto& kv : byS) {
                const std::vector<int>& cur = kv.second;
                size_t s0 = 0;
                if (batches.empty() && (int)cur.size() > first_batch) {
                    Batch b;
                    b.pos.assign(cur.begin(), cur.begin() + first_batch);
                    batches.push_back(b);
                    s0 = first_batch;
                }
                for (size_t s1 = s0; s1 < cur.size(); s1 += max_batch) {
                    Batch b;
                    b.pos.assign(cur.begin() + s1, cur.begin() + std::min(cur.size(), s1 + (size_t)max_batch));
                    batches.push_back(b);
                }
            }
        }
        for (auto& b : batches)
            for (int q : b.pos) b.ks.push_back(members[q]);
        return 0;
    }
    // (GDHIP_BATCH_FIRST_BATCH / GDHIP_BATCH_MAX_BATCH: tuning knobs over the settings' defaults)
    int s_first_batch() const {
        if (s.first_batch > 0) return s.first_batch;
        const char* e = getenv("GDHIP_BATCH_FIRST_BATCH");
        return e && atoi(e) > 0 ? atoi(e) : 128;
    }
    int s_max_batch() const {
        if (s.max_batch > 0) return s.max_batch;
        const char* e = getenv("GDHIP_BATCH_MAX_BATCH");
        return e && atoi(e) > 0 ? atoi(e) : 320;
    }
    double s_max_batch_bytes() const { return s.max_batch_bytes > 0 ? s.max_batch_bytes : 24e9; }
    int s_two_min() const { return s.two_streams_min > 0 ? s.two_streams_min : 64; }
    int s_two_split() const { return s.two_streams_split > 0 ? s.two_streams_split : 400; }
    int s_kopt_split_min() const { return s.kopt_split_min > 0 ? s.kopt_split_min : 256; }

    bool is_side(int F) const { return std::find(side_classes.begin(), side_classes.end(), F) != side_classes.end(); }

    int enqueue_batch(int F, const Batch& b, void* force_ctx) {
        const std::vector<int>& members = classes.at(F);
        void* bctx;
        if (force_ctx)
            bctx = force_ctx;
        else if (!side_classes.empty() || P > s_two_split())
            bctx = conv_ctxs[is_side(F) && conv_ctxs.size() > 1 ? 1 : 0];
        else
            bctx = conv_ctxs[batch_no % conv_ctxs.size()];
        ++batch_no;
        const int B = (int)b.pos.size();
        const int64_t item = (int64_t)F * F * 8;
        void* d_hist;
        {
            std::lock_guard<std::mutex> g(enq_mu);  // (the deferred shear chain may be adding the up-scaled classes' buffers)
            d_hist = hists.at(F);
        }
        int rc = 0;
        bool whole = B == (int)members.size();
        for (int q = 0; q < B && whole; ++q) whole = b.pos[q] == q;
        // (a batch that is not the whole class convolves its histograms where they lie: the kernels take an index list)
        std::vector<int32_t> hidx(b.pos.begin(), b.pos.end());
        std::vector<double> rxb(B), ryb(B), ccb(B);
        std::vector<int32_t> wb(B), fb(B);
        for (int q = 0; q < B; ++q) {
            const int k = b.ks[q];
            rxb[q] = rx[k], ryb[q] = ry[k], ccb[q] = cc[k], wb[q] = (int32_t)winw[k], fb[q] = flags[k];
        }
        void* d_P = pool.take((int64_t)B * item, &rc);
        if (!d_P) return dev_fail(rc, h);
        call_blocks.push_back(d_P);
        int32_t* status = status_pinned + status_at;
        for (int q = 0; q < B; ++q) status[q] = kStatusPending;  // (overwritten by the batch's last launch: see first_part_flag)
        if (grid_off + (int64_t)B * F * F > grids_doubles) return fail(GD_ERR_BADARG, "grids_pinned is too small");
        mark(bctx == h ? "conv: enqueue on main" : "conv: enqueue on twin", B, F);
        GDB_DEV(bctx, ops.density2d_enqueue(bctx, B, F, d_hist, whole ? nullptr : hidx.data(), rxb.data(), ryb.data(), ccb.data(),
                                            wb.data(), fb.data(), s.boundary_correction_order, s.mult_bias_correction_order, d_P,
                                            status));
        mark("conv: enqueued");
        if (s.want_levels && levels) {
            std::vector<double> lv((size_t)B * s.ncontours);
            std::vector<int32_t> ls(B);
            GDB_DEV(bctx, ops.contour_levels(bctx, B, F, d_P, s.contours, s.ncontours, lv.data(), ls.data()));
            for (int q = 0; q < B; ++q) {
                memcpy(levels + (size_t)b.ks[q] * s.ncontours, &lv[(size_t)q * s.ncontours], sizeof(double) * s.ncontours);
                level_status[b.ks[q]] = ls[q];
            }
        }
        GDB_DEV(bctx, ops.d2h_async(bctx, grids + grid_off, d_P, (int64_t)B * item));
        for (int q = 0; q < B; ++q) {
            double* m = M(b.ks[q]);
            m[1] = (double)(grid_off + (int64_t)q * F * F);
            m[29] = bctx == twin ? 1.0 : 0.0;
            m[30] = (double)(status_at + q);
        }
        grid_off += (int64_t)B * F * F;
        status_at += B;
        return 0;
    }

    int run_class(int F, const std::vector<char>& only, void* force_ctx) {
        std::vector<Batch> batches;
        GDB_TRY(class_batches(F, only, batches));
        for (const Batch& b : batches) GDB_TRY(enqueue_batch(F, b, force_ctx));
        return 0;
    }

    // every class, all pairs: classes that go to the second stream are queued there right after the main class's first batch
    int enqueue_all() {
        std::vector<int> main_, side;
        for (int F : order) (is_side(F) ? side : main_).push_back(F);
        const std::vector<char> all;
        std::vector<Batch> first;
        size_t first_done = 0;
        if (!side.empty() && !main_.empty()) {
            GDB_TRY(class_batches(main_[0], all, first));
            if (!first.empty()) {
                GDB_TRY(enqueue_batch(main_[0], first[0], nullptr));
                first_done = 1;
            }
        }
        for (int F : side) GDB_TRY(run_class(F, all, nullptr));
        for (size_t q = 0; q < main_.size(); ++q) {
            if (q == 0 && !side.empty()) {
                for (size_t b = first_done; b < first.size(); ++b) GDB_TRY(enqueue_batch(main_[0], first[b], nullptr));
                continue;
            }
            GDB_TRY(run_class(main_[q], all, nullptr));
        }
        return 0;
    }

    // -- getAutoBandwidth2D for the batch (mcsamples.py:1325-1419): the optimiser's launches, unit conversions,
    //    de-rotation of the sheared kernels, fallbacks, widening; on_chunk(ks, index of the launch, number of launches)
    //    after every launch.
    //    Sequential mode (`staged` false): every launch is one blocking gd_kopt2d call on the main context.
    //    Staged mode (large calls with a second and a third stream): the base grid's pairs are cut into parts of about one
    //    block per CU; the calling thread enqueues stage A of every part (DCT, fixed point, functionals) back to back on the
    //    main stream, a second thread runs stage B (get_h: the serial TNC minimisations) of part k on the third stream --
    //    beside stage A of part k + 1 -- and hands the part's final bandwidths to on_chunk, which enqueues its convolution.
    struct Launch {
        int F = 0;
        std::vector<int> ks;   // plan indices in batch order
        int na = 0;            // leading sheared rows
        void* d_hist = nullptr;  // class buffer
        std::vector<int> pos;  // positions within the class buffer of the non-sheared rows
        bool whole = false;    // the class's buffer as it is
        // staged mode
        void* d_batch = nullptr;
        bool own = false;
        void* d_rows = nullptr;
        int32_t ticket = -1;
        std::vector<double> ne, fb, ci;
        std::vector<int32_t> dc;
    };
    typedef std::function<int(const std::vector<int>&, int, int)> ChunkFn;

    // a launch that holds sheared rows, or belongs to a class the deferred chain bins, waits for that chain here
    template <class BufFn>
    int ready_for(Launch& L, const BufFn& class_buffer) {
        if (L.na > 0 && L.d_hist && !shear_joined.load() && !getenv("GDHIP_BATCH_SHEAR_JOIN_WHOLE")) {
            // the sheared rows are all this launch is waiting for (its own class has its buffer)
            while (shear_hist_state.load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
            mark("shear: histograms awaited");
            if (shear_hist_state.load(std::memory_order_acquire) == 2) {
                (void)join_shear();
                return shear_hist_rc ? shear_hist_rc : GD_ERR_HIP;
            }
        } else if (L.na > 0 || !L.d_hist) {
            const int e = join_shear();
            if (e) return e;
            if (!L.d_hist) L.d_hist = class_buffer(L.F);
            if (!L.d_hist) return fail(GD_ERR_BADARG, "a grid-size class has no histograms");
        }
        return 0;
    }

    int build_batch(Launch& L) {  // the launch's histograms as one device block (a gather unless the class buffer serves)
        const int B = (int)L.ks.size(), F = L.F;
        const int64_t item = (int64_t)F * F * 8;
        L.d_batch = L.d_hist, L.own = false;
        if (!L.whole) {
            int rc = 0;
            L.d_batch = pool.take((int64_t)B * item, &rc);
            if (!L.d_batch) return dev_fail(rc, h);
            L.own = true;
            if (L.na) {
                std::vector<int32_t> idx(L.na);
                for (int a = 0; a < L.na; ++a) idx[a] = a;
                GDB_DEV(h, ops.gather_items(h, L.d_batch, 0, shear.d_rot, idx.data(), L.na, item));
            }
            if (!L.pos.empty()) {
                std::vector<int32_t> idx(L.pos.begin(), L.pos.end());
                GDB_DEV(h, ops.gather_items(h, L.d_batch, L.na, L.d_hist, idx.data(), (int)idx.size(), item));
            }
        }
        L.ne.resize(B), L.fb.resize(B), L.ci.resize(B), L.dc.resize(B);
        for (int row = 0; row < B; ++row) {
            const int k = L.ks[row];
            const bool rowA = row < L.na;
            L.ne[row] = plan.neff[k], L.dc[row] = plan.has_limits[k] ? 0 : 1;
            L.fb[row] = rowA ? -1.0 : plan.fallback_t[k];
            L.ci[row] = rowA ? 0.0 : ps.actual[k];
        }
        return 0;
    }

    // the optimiser's rows of one launch -> bandwidths in parameter units (W), records, fallbacks
    int absorb_rows(const Launch& L, const std::vector<double>& out) {
        const int B = (int)L.ks.size();
        for (int row = 0; row < B; ++row) {
            const double* o = &out[(size_t)row * 12];
            if (o[7] == 0 && o[11] != 0) return fail(GD_ERR_BADARG, "bias not positive definite");  // kde_bandwidth.py:229-230, out of get_h
        }
        for (int row = 0; row < B; ++row) {
            const int k = L.ks[row];
            const double* o = &out[(size_t)row * 12];
            memcpy(M(k) + 6, o, 12 * sizeof(double));
            double hx, hy, c;
            if (row < L.na) {
                // de-rotate the sheared kernel (mcsamples.py:1379-1390): kernelC = S K S^T for the 2 x 2 case
                const double hxa = o[8] * shear.r1s[row], hya = o[9] * shear.r2s[row], ca = o[10];
                const double* S = &plan.S[(size_t)4 * k];
                const double k00 = hxa * hxa, k01 = hxa * hya * ca, k11 = hya * hya;
                const double t00 = S[0] * k00 + S[1] * k01, t01 = S[0] * k01 + S[1] * k11;
                const double t10 = S[2] * k00 + S[3] * k01, t11 = S[2] * k01 + S[3] * k11;
                const double c00 = t00 * S[0] + t01 * S[1], c01 = t00 * S[2] + t01 * S[3], c11 = t10 * S[2] + t11 * S[3];
                const double sx = sqrt(c00), sy = sqrt(c11);
                hx = plan.swap[k] ? sy : sx, hy = plan.swap[k] ? sx : sy, c = c01 / sqrt(c00 * c11);
            } else {
                hx = o[8] * plan.rangex[k], hy = o[9] * plan.rangey[k], c = o[10];
            }
            if (o[7] != 0) {  // "2D fixed point: no root in [0, 0.1]": the fallback widths (mcsamples.py:1402-1409)
                if (s.raise_on_bandwidth_errors) {
                    char buf[200];
                    snprintf(buf, sizeof buf, "2D kernel density bandwidth optimizer failed for pair %d (columns %d, %d). "
                             "Using fallback width: 2D fixed point: no root in [0, 0.1]", k, (int)ps.jx[k], (int)ps.jy[k]);
                    return fail(GD_ERR_SOLVER, buf);
                }
                ps.warn[k] |= 4;
                const double d = py_pow(plan.neff[k], 1.0 / 6);
                hx = par[ps.jx[k]].sigma_range / d, hy = par[ps.jy[k]].sigma_range / d;
                c = std::max(std::min(ps.actual[k], s.max_corr_2D), -s.max_corr_2D);
            }
            W[(size_t)3 * k] = hx, W[(size_t)3 * k + 1] = hy, W[(size_t)3 * k + 2] = c;
        }
        return 0;
    }

    // which parts' convolutions go to the main stream (behind the optimiser's stage A there): the last two fifths
    static bool part_on_main(int index, int nparts) { return nparts > 1 && index >= nparts - std::max(1, 2 * nparts / 5); }

    int bandwidth_2d(bool staged, void* aux, const ChunkFn& on_chunk) {
        const int base_F = s.fine_bins_2D;
        const int m = s.mult_bias_correction_order;
        std::vector<double> widen;
        if (m) {
            widen.resize(P);
            const double e = 1.0 / 6 - 1.0 / (2 + 4 * (1 + m));
            for (int k = 0; k < P; ++k) widen[k] = 1.1 * py_pow(plan.neff[k], e);
        }
        if (!have_shear && !shear_deferred) GDB_TRY(shear_histograms(h));
        std::vector<int> A;  // (= shear.A, which the deferred chain may still be filling)
        for (int k = 0; k < P; ++k)
            if (plan.branch[k] == 0) A.push_back(k);
        const int nA = (int)A.size();
        auto class_buffer = [&](int F) -> void* {  // (the up-scaled classes' buffers arrive with the deferred chain)
            std::lock_guard<std::mutex> g(enq_mu);
            auto it = hists.find(F);
            return it == hists.end() ? nullptr : it->second;
        };
        std::vector<int> waiting;
        for (int k = 0; k < P; ++k)
            if (plan.branch[k] == 1) {  // rule of thumb (mcsamples.py:1391-1395)
                const double c = std::max(std::min(ps.actual[k], s.max_corr_2D), -s.max_corr_2D);
                const double d = py_pow(plan.neff[k], 1.0 / 6);
                W[(size_t)3 * k] = par[ps.jx[k]].sigma_range / d, W[(size_t)3 * k + 1] = par[ps.jy[k]].sigma_range / d, W[(size_t)3 * k + 2] = c;
                waiting.push_back(k);
            }
        std::vector<Launch> launches;
        bool merged = false;
        for (int F : F_list) {
            const std::vector<int>& mem = classes.at(F);
            std::vector<int> pos_C;
            for (int q = 0; q < (int)mem.size(); ++q)
                if (plan.branch[mem[q]] == 2) pos_C.push_back(q);
            if (F == base_F && !pos_C.empty()) {
                // the sheared pairs ride with the first part of the base grid's own pairs
                const size_t total = pos_C.size() + (size_t)nA;
                size_t part = total;
                if (staged && (int)pos_C.size() >= s_kopt_split_min()) part = base_part_size(total, F);
                // (deferred shear chain: the sheared rows lead the LAST part instead of the first)
                const size_t nparts = shear_deferred ? std::max<size_t>(1, (total + part - 1) / part) : 0;
                for (size_t c0 = 0, first = 1, ip = 0; c0 < pos_C.size(); first = 0, ++ip) {
                    Launch L;
                    const bool carries = shear_deferred ? (ip + 1 == nparts || pos_C.size() - c0 <= part) : first != 0;
                    L.F = F, L.na = carries ? nA : 0, L.d_hist = class_buffer(F);
                    const size_t part_q = (ip == 0 && nparts > 1) ? first_part_size(part) : part;
                    size_t take_ = std::min(pos_C.size() - c0, part_q > (size_t)L.na ? part_q - (size_t)L.na : (size_t)1);
                    if (shear_deferred && carries) take_ = pos_C.size() - c0;  // the last part takes what is left
                    // (measured, GDHIP_KOPT_FIRST_PART=256: a first part of 256 pairs, the others unchanged -- with the two-launch
                    // binning delivered 27.2-27.6 against 27.3-27.7 ms, the stream of triangles 21.0-21.2 against 20.6-20.7: the
                    // result copies saturate PCIe from the first batch on and the copy engine then waits for the second part)
                    L.pos.assign(pos_C.begin() + c0, pos_C.begin() + c0 + take_);
                    c0 += take_;
                    L.whole = L.na == 0 && L.pos.size() == mem.size();
                    for (int q = 0; q < L.na; ++q) L.ks.push_back(A[q]);
                    for (int q : L.pos) L.ks.push_back(mem[q]);
                    launches.push_back(std::move(L));
                }
                merged = true;
                continue;
            }
            if (pos_C.empty()) continue;
            Launch L;
            L.F = F, L.na = 0, L.d_hist = class_buffer(F), L.pos = pos_C, L.whole = pos_C.size() == mem.size();
            for (int q : pos_C) L.ks.push_back(mem[q]);
            launches.push_back(std::move(L));
        }
        if (nA && !merged) {
            GDB_TRY(join_shear(true));
            Launch L;
            L.F = base_F, L.na = nA, L.d_hist = shear.d_rot, L.whole = true;
            L.ks = A;
            launches.push_back(std::move(L));
        }
        const int nl = (int)launches.size();
        // (the deferred shear chain waits for the first part's bandwidths -- unless that part itself waits for the chain)
        if (!staged || nl == 0 || launches[0].na > 0 || !launches[0].d_hist) first_part_done.store(1);
        auto report = [&](std::vector<int> ks, int index) -> int {
            if (!waiting.empty()) {
                // the rule-of-thumb pairs ride with the first report -- except, with the deferred shear chain, those of a
                // class that chain bins (an up-scaled grid): they wait for the last report, when the chain has been joined
                std::vector<int> all, later;
                for (int k : waiting) ((shear_deferred && ps.F[k] != base_F && index + 1 < nl) ? later : all).push_back(k);
                all.insert(all.end(), ks.begin(), ks.end());
                ks.swap(all);
                waiting.swap(later);
            }
            if (m)
                for (int k : ks) W[(size_t)3 * k] *= widen[k], W[(size_t)3 * k + 1] *= widen[k];
            if (shear_deferred && index + 1 >= nl) {
                // the last report convolves the rule-of-thumb pairs of the classes the chain bins: the WHOLE chain, not only
                // its sheared histograms (ready_for), has to be through
                const int e = join_shear();
                if (e) return e;
            }
            if (!ks.empty() || index + 1 >= nl) return on_chunk(ks, index, nl);
            return 0;
        };
        int rc = 0;
        // a launch of the base class whose rows the FIRST binning launch has filled may go ahead of the second
        auto rows_binned_first = [&](const Launch& L) {
            if (bins_first_rows <= 0 || L.F != base_F || L.na > 0 || !L.d_hist) return false;
            for (int q : L.pos)
                if (q >= bins_first_rows) return false;
            return true;
        };
        auto whole_binning_for = [&](Launch& L) -> int {
            const int e = join_binning();
            if (e) return e;
            if (L.F == base_F && !L.pos.empty())  // (a class buffer whose second launch failed has been replaced)
                if (void* p = class_buffer(L.F)) L.d_hist = p;
            return 0;
        };
        if (!staged || nl == 0) {
            rc = join_binning();
            for (int q = 0; q < nl && !rc; ++q) {
                Launch& L = launches[q];
                const int B = (int)L.ks.size();
                rc = whole_binning_for(L);
                if (!rc) rc = ready_for(L, class_buffer);
                if (!rc) rc = build_batch(L);
                std::vector<double> out((size_t)B * 12);
                if (!rc) {
                    mark("kopt: launch", B, L.F);
                    rc = ops.kopt2d(h, B, L.F, L.d_batch, L.ne.data(), L.dc.data(), L.fb.data(), L.ci.data(), out.data());
                    mark("kopt: done");
                    if (rc) dev_fail(rc, h);
                }
                if (L.own) pool.give(L.d_batch);  // (the entry point has waited for its kernels)
                if (!rc) rc = absorb_rows(L, out);
                if (!rc) rc = report(L.ks, q);
            }
            if (!rc && nl == 0) rc = report({}, 0);
        } else {
            // ---- staged: stage A of every launch from this thread, stage B + the hand-over from a second one
            std::mutex mu;
            std::condition_variable cv;
            int n_staged = 0;
            bool abort_ = false;
            // Hand-over of finished parts (round 5): the finisher only runs stage B; a third thread enqueues a part's
            // convolution.  When the finisher did both, the get_h launch of part k + 1 waited for the ~2 ms of host work
            // that enqueueing part k's ~120 convolution launches takes (trace: stage A of part 2 done at 18.9 ms, its get_h
            // started at 22.9), and the convolutions of the later parts were launch-bound behind it.
            std::deque<int> handed;  // parts whose bandwidths are final, in order; -1: no more
            std::future<int> enqueuer = std::async(std::launch::async, [&]() -> int {
                ops.bind_thread(h);
                int e = 0;
                for (;;) {
                    int q;
                    {
                        std::unique_lock<std::mutex> g(mu);
                        cv.wait(g, [&] { return !handed.empty(); });
                        q = handed.front();
                        handed.pop_front();
                    }
                    if (q < 0) return e;
                    if (e) continue;
                    if (part_on_main(q, nl) || !shear_deferred) {
                        // this part's convolution uses the main context: the staging thread must be done with it.  (The
                        // earlier parts go to the second context only -- with the deferred shear chain the staging thread
                        // may still be waiting for that chain when the first part's bandwidths are final, and the first
                        // part's grids are what the result copies start with.)
                        std::unique_lock<std::mutex> g(mu);
                        cv.wait(g, [&] { return n_staged >= nl || abort_; });
                        if (n_staged < nl) {
                            e = GD_ERR_HIP;  // (the staging thread gave up: its error is the call's)
                            continue;
                        }
                    }
                    e = join_binning();  // (long done by now: a report may convolve any row of the class)
                    if (!e) e = report(launches[q].ks, q);
                    if (reports_enqueued++ == 0 && status_at > 0 && !s.results_in_flight && !getenv("GDHIP_BATCH_SHEAR_ON_ENQUEUE"))
                        first_part_flag.store(status_pinned + status_at - 1);  // the last word of the first report's last batch
                    first_part_done.store(1);  // the first part's convolution is enqueued: the deferred shear chain may start
                }
            });
            std::future<int> finisher = std::async(std::launch::async, [&]() -> int {
                ops.bind_thread(aux);
                int e = 0;
                auto hand = [&](int q) {
                    std::lock_guard<std::mutex> g(mu);
                    handed.push_back(q);
                    cv.notify_all();
                };
                for (int q = 0; q < nl; ++q) {
                    {
                        std::unique_lock<std::mutex> g(mu);
                        cv.wait(g, [&] { return n_staged > q || abort_; });
                        if (n_staged <= q) break;  // (the staging thread gave up)
                    }
                    Launch& L = launches[q];
                    const int B = (int)L.ks.size();
                    std::vector<double> out((size_t)B * 12);
                    if (!e) {
                        e = ops.kopt2d_finish(aux, h, L.ticket, B, L.d_rows, out.data());
                        mark("kopt: part finished", q, B);
                        if (e) dev_fail(e, aux);
                    } else {
                        ops.copy_sync(h);  // (never reached in a healthy run: blocks are not freed under running kernels)
                    }
                    // stage B has waited for stage A: the part's input blocks are free
                    if (L.own) pool.give(L.d_batch);
                    pool.give(L.d_rows);
                    if (!e) e = absorb_rows(L, out);
                    if (e || getenv("GDHIP_BATCH_SHEAR_AFTER_GET_H")) first_part_done.store(1);  // (A/B: release on the bandwidths already)
                    if (!e) hand(q);
                }
                first_part_done.store(1);
                hand(-1);
                return e;
            });
            for (int q = 0; q < nl && !rc; ++q) {
                Launch& L = launches[q];
                const int B = (int)L.ks.size();
                if (!(q == 0 && rows_binned_first(L))) rc = whole_binning_for(L);
                if (!rc) rc = ready_for(L, class_buffer);
                if (!rc) rc = build_batch(L);
                if (!rc) {
                    L.d_rows = pool.take((int64_t)B * GD_KOPT_BLOCK_DOUBLES * 8, &rc);
                    if (!L.d_rows) dev_fail(rc, h);
                }
                if (!rc) {
                    mark("kopt: stage A enqueue", q, B);
                    rc = ops.kopt2d_enqueue(h, B, L.F, L.d_batch, L.ne.data(), L.dc.data(), L.fb.data(), L.ci.data(), L.d_rows, &L.ticket);
                    if (rc) dev_fail(rc, h);
                }
                std::lock_guard<std::mutex> g(mu);
                if (rc) {
                    abort_ = true;
                    if (L.own && L.d_batch) ops.copy_sync(h), pool.give(L.d_batch);
                    if (L.d_rows) pool.give(L.d_rows);
                } else {
                    ++n_staged;
                }
                cv.notify_all();
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (n_staged < nl) abort_ = true;
                cv.notify_all();
            }
            mark("kopt: every stage A enqueued");
            const int e = finisher.get();
            const int e2 = enqueuer.get();
            if (!rc) rc = e;
            if (!rc) rc = e2;
        }
        {
            const int e0 = join_binning();
            const int e = join_shear(true);  // (an error path may get here before any launch asked for it)
            if (!rc) rc = e0;
            if (!rc) rc = e;
        }
        if (shear.d_rot) pool.give(shear.d_rot), shear.d_rot = nullptr;
        return rc;
    }

    int run(int32_t* tokens_out2) {
        tokens_out2[0] = tokens_out2[1] = -1;
        const int base_F = s.fine_bins_2D, bco = s.boundary_correction_order;
        const double ss = s.smooth_scale_2D;
        if (fabs(s.max_corr_2D) > 1) return fail(GD_ERR_BADARG, "max_corr_2D cannot be >=1");
        if (bco > 1) return fail(GD_ERR_BADARG, "unknown boundary_correction_order (expected 0 or 1)");
        if (base_F < 8 || base_F > 4096) return fail(GD_ERR_BADARG, "fine_bins_2D out of range");
        bool exchanged = false;
        if (P == 0) {  // (a rank without pairs still takes part in the exchange)
            GDB_TRY(neff_exchange(&exchanged));
            return 0;
        }
        mark("call: start", P);
        int64_t ncols = 0;
        GDB_DEV(h, ops.num_rows(h, &N, &ncols));
        if (ncols < n) return fail(GD_ERR_BADARG, "more parameters than resident columns");
        int32_t hw = 0;
        GDB_DEV(h, ops.weights_kind(h, &hw));
        unit_weights = hw == 0;
        byte_index_weights = hw == 2 && !getenv("GDHIP_NO_WSORT");
        for (int k = 0; k < P; ++k)
            if (pairs[2 * k] < 0 || pairs[2 * k] >= n || pairs[2 * k + 1] < 0 || pairs[2 * k + 1] >= n)
                return fail(GD_ERR_BADARG, "pair index out of range");
        {  // columns in order of first appearance
            std::vector<char> seen(n, 0);
            for (int k = 0; k < 2 * P; ++k)
                if (!seen[pairs[k]]) seen[pairs[k]] = 1, used.push_back(pairs[k]);
        }
        // settings->bandwidths: the caller's (hx, hy, corr) per pair instead of getAutoBandwidth2D (tests inject the
        // oracle's triples); settings->pair_neff: the caller's effective sample number per pair (use_effective_samples_2D)
        const bool injected = ss < 0 && s.bandwidths != nullptr;
        const bool auto_bw = ss < 0 && !injected;
        bool need_neff = false;
        if (auto_bw && !s.pair_neff)
            for (int j : used) need_neff = need_neff || isnan(par[j].neff);
        // Large calls keep three streams busy from the start: the N_eff kernels (fp64 exp-bound) on the main context, the
        // byte-index binning (LDS atomics) on the second, the sheared min/max + re-binning (HBM-bound) on a third the
        // library creates for itself; the per-pair scalars and the branch plan are worked out here meanwhile.
        void* aux = nullptr;
        const bool overlap = auto_bw && P >= 64 && twin != nullptr;
        if (overlap) {
            if (!st.aux) GDB_DEV(h, ops.create_aux(h, &st.aux));
            aux = st.aux;
        }
        std::future<int> neff_f, shear_f;
        bin_launching.store(0);
        hold_neff_for_binning = overlap && need_neff && unit_weights && !getenv("GDHIP_BATCH_NO_NEFF_HOLD");
        if (overlap && need_neff)
            neff_f = std::async(std::launch::async, [this] {
                ops.bind_thread(h);
                return neff_batch(used, true);
            });
        bmin.assign(n, NAN), bmax.assign(n, NAN);
        for (int j : used) bin_edges(par[j], &bmin[j], &bmax[j]);
        pair_scalars(s, n, corrmat, pairs, P, ps);
        fwx.resize(P), fwy.resize(P);
        for (int k = 0; k < P; ++k) {
            const int F = ps.F[k];
            if (!classes.count(F)) F_list.push_back(F);
            classes[F].push_back(k);
            fwx[k] = (bmax[ps.jx[k]] - bmin[ps.jx[k]]) / (F - 1);
            fwy[k] = (bmax[ps.jy[k]] - bmin[ps.jy[k]]) / (F - 1);
            double* m = M(k);
            for (int q = 0; q < GD_BATCH2D_META; ++q) m[q] = NAN;
            m[0] = F, m[5] = -1, m[23] = bmin[ps.jx[k]], m[24] = bmax[ps.jx[k]], m[25] = bmin[ps.jy[k]], m[26] = bmax[ps.jy[k]];
            m[27] = ps.corr[k], m[28] = (double)ps.nbin2D[k];
        }
        int64_t need = 0;
        for (int k = 0; k < P; ++k) need += (int64_t)ps.F[k] * ps.F[k];
        if (need > grids_doubles && !s.bandwidths_only) {
            if (neff_f.valid()) neff_f.get();
            return fail(GD_ERR_BADARG, "grids_pinned is too small");
        }
        std::vector<double> rngx(P), rngy(P);
        for (int k = 0; k < P; ++k) rngx[k] = bmax[ps.jx[k]] - bmin[ps.jx[k]], rngy[k] = bmax[ps.jy[k]] - bmin[ps.jy[k]];
        int rc = 0;
        if (auto_bw) {
            if (overlap) {
                const bool split_classes = F_list.size() > 1;
                bin_done.store(0);
                first_part_done.store(0);
                first_part_flag.store(nullptr), reports_enqueued = 0;
                {
                    // deferred shear chain (see shear_future): large unit-weight calls whose main class takes the byte-index
                    // route; GDHIP_BATCH_SHEAR_DEFERRED=0 restores the join in front of the optimiser
                    const char* e = getenv("GDHIP_BATCH_SHEAR_DEFERRED");
                    const char* emin = getenv("GDHIP_BATCH_SHEAR_DEFERRED_MIN");  // (tests lower it)
                    shear_deferred = (e ? atoi(e) != 0 : true) && unit_weights && P >= (emin ? atoi(emin) : 256);
                    if (shear_deferred && !st.aux2) {
                        // (measured: the chain's stream at the LOWEST priority, so that the optimiser's blocks would be
                        // dispatched ahead of it -- 30.0-30.4 ms per delivered triangle against 29.5-29.7 with the high
                        // priority create_aux gives it, 30.6-30.8 with the chain joined in front of the optimiser: its
                        // 1024-thread blocks hold their CUs either way, and a low priority only delays the chain's end)
                        int rc2 = ops.create_aux(h, &st.aux2);
                        if (!rc2 && ops.stream_priority && getenv("GDHIP_BATCH_SHEAR_LOW_PRIORITY")) rc2 = ops.stream_priority(st.aux2, -1);
                        if (rc2) {
                            if (st.aux2) ops.destroy_aux(st.aux2), st.aux2 = nullptr;
                            shear_deferred = false, (void)ops.last_error(h);
                        }
                    }
                    shear_after_binning = shear_deferred;
                }
                // (the branch plan needs the limits and the covariance only)
                rc = make_plan(s, par, n, cov, ps, rngx, rngy, 0.2, plan, err);
                bins_first_rows = 0;
                bins_first_done.store(0);
                const bool will_stage = aux != nullptr && !s.want_levels && P >= s_two_min() && P > s_two_split() && !s.bandwidths_only;
                if (!rc && shear_deferred && will_stage && base_F == 256 && main_class() == 256 && !getenv("GDHIP_BATCH_ONE_BINNING")) {
                    std::vector<int>& mem = classes[256];
                    std::stable_partition(mem.begin(), mem.end(), [&](int k) { return plan.branch[k] == 2; });
                    size_t nC = 0, nA = 0;
                    for (int k : mem) nC += plan.branch[k] == 2;
                    for (int k = 0; k < P; ++k) nA += plan.branch[k] == 0;
                    if ((int)nC >= s_kopt_split_min()) {
                        const size_t part = base_part_size(nC + nA, 256), first = std::min(nC, first_part_size(part));
                        if (nC + nA > part && first < mem.size()) bins_first_rows = (int)first;
                    }
                }
                bin_joined = false, bin_rc = 0;
                bin_future = std::async(std::launch::async, [this, split_classes] {
                                 ops.bind_thread(twin);
                                 const int e = binning(twin, split_classes ? 1 : 0);
                                 bins_first_done.store(1);
                                 bin_done.store(1);
                                 return e;
                             }).share();
                if (!rc) shear_hist_state.store(0);
                if (!rc)  // the shear chain starts at once
                    shear_f = std::async(std::launch::async, [this, aux, split_classes] {
                        void* sctx = shear_deferred ? st.aux2 : aux;
                        ops.bind_thread(sctx);
                        if (shear_after_binning) {
                            const auto t0 = std::chrono::steady_clock::now();
                            while (!bin_done.load() && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50))
                                std::this_thread::sleep_for(std::chrono::microseconds(50));
                            if (!getenv("GDHIP_BATCH_SHEAR_AT_ONCE")) {
                                const auto t1 = std::chrono::steady_clock::now();
                                while (!first_part_done.load() && std::chrono::steady_clock::now() - t1 < std::chrono::milliseconds(20))
                                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                            }
                            // (GDHIP_BATCH_SHEAR_HOLD_US: the chain's start held back further -- tuning knob)
                            if (const volatile int32_t* flag = first_part_flag.load()) {  // ... and has run (first_part_flag)
                                const auto t2 = std::chrono::steady_clock::now();
                                while (*flag == kStatusPending && std::chrono::steady_clock::now() - t2 < std::chrono::milliseconds(20))
                                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                            }
                            if (first_part_done.load() && shear_hold_us() > 0)
                                std::this_thread::sleep_for(std::chrono::microseconds(shear_hold_us()));
                            mark("shear: main binning has run, first part's bandwidths final");
                        }
                        const int e = shear_histograms(sctx);
                        shear_hist_rc = e;
                        shear_hist_state.store(e ? 2 : 1, std::memory_order_release);
                        if (const char* d = getenv("GDHIP_BATCH_TEST_SIDE_BINNING_DELAY_MS"))  // test hook: a slow second half
                            std::this_thread::sleep_for(std::chrono::milliseconds(atoi(d)));
                        const int e2 = split_classes ? binning(sctx, 2) : 0;  // the up-scaled classes, behind the shear chain
                        return e ? e : e2;
                    });  // (no plan: the call fails; the side classes are not needed)
                int e = neff_f.valid() ? neff_f.get() : 0;
                if (!rc) rc = e;
                if (!rc) rc = neff_complete(&exchanged);  // (multi-rank: the other ranks' values, from this thread)
                if (!rc) fill_plan(par, ps, plan, s.pair_neff);
                if (shear_f.valid()) {
                    if (shear_deferred) {
                        shear_future = std::move(shear_f), shear_joined.store(false);  // joined by the first launch that needs it
                    } else {
                        e = shear_f.get();
                        if (!rc) rc = e;
                    }
                }
                if (bins_first_rows > 0 && !rc) {
                    // the first optimiser part's rows only; whoever needs more joins the binning (bandwidth_2d)
                    while (!bins_first_done.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
                    mark("binning: first part's rows done", bins_first_rows);
                } else {
                    e = join_binning();
                    if (!rc) rc = e;
                }
                if (rc) (void)join_shear(true);
            } else {
                rc = s.pair_neff ? 0 : neff_batch(used, true);
                if (!rc) rc = neff_complete(&exchanged);
                if (!rc) rc = binning(h);
                if (!rc) rc = make_plan(s, par, n, cov, ps, rngx, rngy, 0.2, plan, err);
                if (!rc) fill_plan(par, ps, plan, s.pair_neff);
            }
        } else {
            rc = neff_exchange(&exchanged);  // the collective is unconditional: once per call on every rank
            if (!rc && !s.bandwidths_only) rc = binning(h);
        }
        mark("binning / N_eff / plan joined");
        if (rc) return cleanup(rc);
        // ---- convolution set-up: flag bits (mcsamples.py:1688-1703, 1794): bits 0/1 = x bot/top, 2/3 = y bot/top,
        //      4/5 = x/y periodic, 6 = has_prior
        flags.resize(P), group.resize(P);
        for (int k = 0; k < P; ++k) {
            const gd_param2d &px = par[ps.jx[k]], &py = par[ps.jy[k]];
            const int lbx = px.periodic ? 0 : (px.has_limits_bot ? 1 : 0) | (px.has_limits_top ? 2 : 0);
            const int lby = py.periodic ? 0 : (py.has_limits_bot ? 1 : 0) | (py.has_limits_top ? 2 : 0);
            const int has_prior = has_limits(px) || has_limits(py);
            flags[k] = lbx | ((px.periodic ? 1 : 0) << 4) | (lby << 2) | ((py.periodic ? 1 : 0) << 5) | (has_prior << 6);
            group[k] = (flags[k] & 48) * 2 + ((has_prior && bco >= 0) ? 1 : 0);
        }
        rx.assign(P, NAN), ry.assign(P, NAN), cc.assign(P, NAN), smooth.assign(P, NAN), winw.assign(P, 0), W.assign((size_t)3 * P, NAN);
        const bool lazy = !s.want_levels;
        conv_ctxs = {h};
        if (lazy && twin && P >= s_two_min()) {
            if (P > s_two_split()) {
                for (int F : F_list)
                    if ((int)classes.at(F).size() < 64) side_classes.push_back(F);
                if (side_classes.size() == F_list.size()) side_classes.clear();
            }
            conv_ctxs.push_back(twin);
        }
        // largest class (in bytes) first: the copy of the last, smallest one is the only exposed one
        order = F_list;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            return (int64_t)classes.at(a).size() * a * a > (int64_t)classes.at(b).size() * b * b;
        });
        std::vector<int> all_k(P);
        for (int k = 0; k < P; ++k) all_k[k] = k;
        if (auto_bw) {
            const bool staged = aux != nullptr && conv_ctxs.size() > 1 && P > s_two_split() && !s.bandwidths_only;
            if (staged) {
                // every part's convolution is enqueued (by the thread that finishes the parts) as soon as its bandwidths are
                // final: the first parts on the second stream, beside the optimiser's stage A of the later parts on the
                // main stream; the last parts behind stage A on the main stream, so that both streams end together
                ChunkFn on_chunk = [&](const std::vector<int>& ks, int index, int nparts) -> int {
                    set_scales(ks);
                    std::vector<char> only(P, 0);
                    for (int k : ks) only[k] = 1;
                    const bool to_main = part_on_main(index, nparts);
                    // (measured alternatives, C3 step: every part on the second stream 31.4 ms, the last parts on the third
                    // -- high-priority -- stream 36.3 ms, against 30.4 ms as below)
                    for (int F : order) {
                        void* target = (is_side(F) || !to_main) ? conv_ctxs[1] : conv_ctxs[0];
                        GDB_TRY(run_class(F, only, target));
                    }
                    return 0;
                };
                rc = bandwidth_2d(true, aux, on_chunk);
            } else {
                ChunkFn on_chunk = [&](const std::vector<int>&, int, int) -> int { return 0; };
                rc = bandwidth_2d(false, nullptr, on_chunk);
                if (!rc) {
                    set_scales(all_k);
                    if (!s.bandwidths_only) rc = enqueue_all();
                }
            }
            for (int k = 0; k < P && !rc; ++k) {
                double* m = M(k);
                m[2] = W[(size_t)3 * k], m[3] = W[(size_t)3 * k + 1], m[4] = W[(size_t)3 * k + 2];
                m[5] = plan.branch[k], m[31] = plan.neff[k];
            }
        } else if (injected) {
            for (int k = 0; k < P; ++k) {
                for (int q = 0; q < 3; ++q) W[(size_t)3 * k + q] = s.bandwidths[(size_t)3 * k + q];
                double* m = M(k);
                m[2] = W[(size_t)3 * k], m[3] = W[(size_t)3 * k + 1], m[4] = W[(size_t)3 * k + 2];
            }
            set_scales(all_k);
            if (!s.bandwidths_only) rc = enqueue_all();
        } else {
            for (int k = 0; k < P; ++k) {
                if (ss < 1.0) {
                    rx[k] = ss * par[ps.jx[k]].err / fwx[k];
                    ry[k] = ss * par[ps.jy[k]].err / fwy[k];
                } else {
                    rx[k] = ry[k] = ss * ps.F[k] / (double)ps.nbin2D[k];
                }
                cc[k] = ps.corr[k];
                finish_scale(k);
            }
            if (!s.bandwidths_only) rc = enqueue_all();
        }
        if (rc) return cleanup(rc);
        for (int k = 0; k < P; ++k) {
            double* m = M(k);
            m[18] = rx[k], m[19] = ry[k], m[20] = cc[k], m[21] = (double)winw[k], m[22] = ps.warn[k];
        }
        mark("all batches enqueued");
        for (auto& kv : hists) call_blocks.push_back(kv.second);
        hists.clear();
        for (void* p : bin_stale) call_blocks.push_back(p);
        bin_stale.clear();
        GDB_DEV(h, ops.copy_mark(h, &tokens_out2[0]));
        if (conv_ctxs.size() > 1) GDB_DEV(twin, ops.copy_mark(twin, &tokens_out2[1]));
        if (!lazy) {
            for (void* c : conv_ctxs) GDB_DEV(c, ops.copy_sync(c));
            for (void* p : call_blocks) pool.give(p);
            call_blocks.clear();
            return 0;
        }
        // This call's blocks wait for its copies.  The previous call's copies are ahead of this call's on the copy
        // streams: completing it here costs no waiting, and its device blocks return to the pool even if nobody ever
        // read its grids.
        Pending cur;
        cur.blocks.swap(call_blocks);
        cur.tok_main = tokens_out2[0], cur.tok_twin = tokens_out2[1], cur.twin = conv_ctxs.size() > 1 ? twin : nullptr;
        cur.live = true;
        const int e_prev = complete_pending(st, ops, h, st.prev);
        st.prev = std::move(cur);
        mark("previous call completed; return");
        dump_timeline();
        if (e_prev) return dev_fail(e_prev, h);
        return 0;
    }

    // an error after device work was started: wait for whatever is in flight, hand every block back
    int cleanup(int rc) {
        (void)join_binning();
        (void)join_shear(true);  // (a deferred chain still running on its own context uses the pool and the class table)
        ops.copy_sync(h);
        if (twin) ops.copy_sync(twin);
        for (auto& kv : hists) pool.give(kv.second);
        hists.clear();
        for (void* p : bin_stale) pool.give(p);
        bin_stale.clear();
        if (shear.d_rot) pool.give(shear.d_rot), shear.d_rot = nullptr;
        for (void* p : call_blocks) pool.give(p);
        call_blocks.clear();
        return rc;
    }
#undef GDB_DEV
#undef GDB_TRY
};

}  // namespace gdb
