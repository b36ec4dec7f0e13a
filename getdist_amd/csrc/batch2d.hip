// gd_density2d_batch: the native entry for a batch of parameter pairs.  The plan and the choreography live in
// batch2d.hpp (host C++, shared with the CPU test harness); this file binds its table of device entry points to the
// library's own C ABI and keeps the per-context state (block pool, cached index columns, the last call in flight).
#include "batch1d.hpp"
#include "batch2d.hpp"
#include "ctx.hpp"

namespace {

gd_ctx* C(void* h) { return (gd_ctx*)h; }

const gdb::Ops kOps = {
    /* bind_thread */ [](void* h) { return gd_bind_thread(C(h)); },
    /* num_rows */ [](void* h, int64_t* N, int64_t* n) { return gd_num_rows(C(h), N, n); },
    /* weights_kind */
    [](void* h, int32_t* hw) {  // 0: unit weights, 1: weights, 2: real (non-integral) weights with a known total
        gd_ctx* c = C(h);
        *hw = c->w == nullptr ? 0 : ((!c->w8 && !c->w_integral && c->w_sum > 1e-200 && c->w_sum < 1e200) ? 2 : 1);
        return 0;
    },
    /* dev_alloc */ [](void* h, int64_t bytes, void** out) { return gd_dev_alloc(C(h), bytes, out); },
    /* dev_free */ [](void* h, void* p) { return gd_dev_free(C(h), p); },
    /* prebin8_batch */
    [](void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F, void* const* d_idx,
       int64_t* bad) { return gd_prebin8_batch(C(h), cols, ncols, binmin, width, F, d_idx, bad); },
    /* hist2d_prebinned8 */
    [](void* h, int32_t B, const void* const* ix, const void* const* iy, void* d_hist) {
        return gd_hist2d_prebinned8(C(h), B, ix, iy, d_hist);
    },
    /* prebin8_hist2d */
    [](void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, void* const* d_idx, int64_t* bad,
       int32_t B, const void* const* ix, const void* const* iy, void* d_hist) {
        return gd_prebin8_hist2d(C(h), cols, ncols, binmin, width, d_idx, bad, B, ix, iy, d_hist);
    },
    /* prebin */
    [](void* h, int32_t col, double binmin, double width, int32_t F, void* d_idx) {
        return gd_prebin(C(h), col, binmin, width, F, d_idx);
    },
    /* hist2d_prebinned */
    [](void* h, int32_t B, const void* const* ix, const void* const* iy, int32_t F, void* d_hist) {
        return gd_hist2d_prebinned(C(h), B, ix, iy, F, d_hist);
    },
    /* minmax_affine */
    [](void* h, int32_t B, const int32_t* ci, const int32_t* cj, const double* a, const double* b, double* out) {
        return gd_minmax_affine(C(h), B, ci, cj, a, b, out);
    },
    /* hist2d_sheared */
    [](void* h, int32_t B, const int32_t* ci, const int32_t* cj, const double* r0, const double* r1, const double* xmin,
       const double* dx, const double* ymin, const double* dy, int32_t F, void* d_hist) {
        return gd_hist2d_sheared(C(h), B, ci, cj, r0, r1, xmin, dx, ymin, dy, F, d_hist);
    },
    /* kopt2d */
    [](void* h, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr, const double* fallback_t,
       const double* corr, double* out) { return gd_kopt2d(C(h), B, F, d_hist, neff, do_corr, fallback_t, corr, out); },
    /* gather_items */
    [](void* h, void* d_dst, int64_t dst_first, const void* d_src, const int32_t* index, int32_t count, int64_t item_bytes) {
        return gd_gather_items(C(h), (char*)d_dst + dst_first * item_bytes, d_src, index, count, item_bytes);
    },
    /* density2d_enqueue */
    [](void* h, int32_t B, int32_t F, const void* d_hist, const int32_t* hist_index, const double* rx, const double* ry,
       const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc, void* d_P, int32_t* status_pinned) {
        return gd_density2d_enqueue_indexed(C(h), B, F, d_hist, hist_index, rx, ry, corr, winw, flags, bco, mbc, d_P, status_pinned);
    },
    /* d2h_async */ [](void* h, void* dst, const void* d_src, int64_t bytes) { return gd_memcpy_d2h_async(C(h), dst, d_src, bytes); },
    /* copy_mark */ [](void* h, int32_t* token) { return gd_copy_mark(C(h), token); },
    /* copy_wait: a context destroyed meanwhile has synchronised its copy stream on the way out */
    [](void* h, int32_t token) {
        if (!gd_ctx_alive(C(h))) return 0;
        const int rc = gd_copy_wait(C(h), token);
        return rc == GD_ERR_BADARG ? 0 : rc;  // (a new context at the old address: no such mark)
    },
    /* copy_sync */ [](void* h) { return gd_ctx_alive(C(h)) ? gd_copy_sync(C(h)) : 0; },
    /* contour_levels */
    [](void* h, int32_t B, int32_t F, const void* d_P, const double* contours, int32_t nc, double* out, int32_t* status) {
        return gd_contour_levels(C(h), B, F, d_P, contours, nc, out, status);
    },
    /* autocov_lags_batch */
    [](void* h, const int32_t* cols, int32_t ncols, const double* means, int64_t k0, int32_t nlags, double* out) {
        return gd_autocov_lags_batch(C(h), cols, ncols, means, k0, nlags, out);
    },
    /* kde_lag_sums_batch */
    [](void* h, const int32_t* cols, int32_t ncols, const double* inv4s2, const int64_t* lags, int32_t nlags, double* out) {
        return gd_kde_lag_sums_batch(C(h), cols, ncols, inv4s2, lags, nlags, out);
    },
    /* kde_lag_sums */
    [](void* h, int32_t col, double inv4s2, const int64_t* lags, int32_t nlags, double* out) {
        return gd_kde_lag_sums(C(h), col, inv4s2, lags, nlags, out);
    },
    /* last_error */ [](void* h) { return gd_last_error(C(h)); },
    /* create_aux */
    [](void* h, void** aux) {
        gd_ctx* a = nullptr;
        int rc = gd_create(C(h)->device, &a);
        if (rc == 0) rc = gd_attach_samples(a, C(h));
        if (rc == 0) rc = gd_stream_priority(a, 1);  // small latency-critical kernels (get_h) beside saturating launches
        if (rc) {
            if (a) gd_destroy(a);
            return gd_fail(C(h), rc, "could not create the third context of a batched call");
        }
        *aux = a;
        return 0;
    },
    /* destroy_aux */
    [](void* aux) {
        gd_destroy(C(aux));
        return 0;
    },
    /* kopt2d_enqueue */
    [](void* h, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr, const double* fallback_t,
       const double* corr, void* d_rows, int32_t* ticket) {
        return gd_kopt2d_enqueue(C(h), B, F, d_hist, neff, do_corr, fallback_t, corr, d_rows, ticket);
    },
    /* kopt2d_finish */
    [](void* h, void* stage_a_h, int32_t ticket, int32_t B, void* d_rows, double* out) {
        return gd_kopt2d_finish(C(h), C(stage_a_h), ticket, B, d_rows, out);
    },
    /* comm_world */ [](void* h) { return C(h)->comm ? C(h)->comm_world : 0; },
    /* comm_allreduce_sum */ [](void* h, double* inout, int64_t count) { return gd_comm_allreduce_sum(C(h), inout, count); },
    /* stream_priority */ [](void* ctx, int level) { return gd_stream_priority(C(ctx), level); },
    /* prebin_batch */
    [](void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F, void* const* d_idx) {
        return gd_prebin_batch(C(h), cols, ncols, binmin, width, F, d_idx);
    },
};

const gdb::Ops1D kOps1D = {
    /* hist1d_dev */
    [](void* h, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F, void* d_hist) {
        return gd_hist1d_dev(C(h), cols, ncols, binmin, width, F, d_hist);
    },
    /* isj1d_dev */
    [](void* h, int32_t B, int32_t F, const void* d_hist, const double* neff, double* hfrac, int32_t* status) {
        return gd_isj1d_dev(C(h), B, F, d_hist, neff, hfrac, status);
    },
    /* density1d_dev */
    [](void* h, int32_t B, int32_t F, const void* d_hist, const double* smooth, const int32_t* winw, const int32_t* flags,
       int32_t bco, int32_t mbc, double* P_out, int32_t* status) {
        return gd_density1d_dev(C(h), B, F, d_hist, smooth, winw, flags, bco, mbc, P_out, status);
    },
    /* fetch */
    [](void* h, void* dst, const void* d_src, int64_t bytes) {
        const int rc = gd_fetch(C(h), dst, d_src, (size_t)bytes);
        return rc ? rc : gd_stream_sync(C(h));
    },
};

void release_state(gd_ctx* ctx, bool destroy) {
    gdb::State* st = (gdb::State*)ctx->batch_state;
    if (!st) return;
    gdb::release_all(*st, kOps, ctx);
    if (destroy) {
        delete st;
        ctx->batch_state = nullptr;
    }
}

gdb::State& state_of(gd_ctx* ctx) {
    if (!ctx->batch_state) {
        ctx->batch_state = new gdb::State();
        ctx->batch_state_release = release_state;
    }
    return *(gdb::State*)ctx->batch_state;
}

}  // namespace

extern "C" {

int gd_batch2d_grid_sizes(const gd_batch2d_settings* settings, int32_t n, const double* corr, const int32_t* pairs, int32_t P,
                          int32_t* F_out) {
    if (!settings || !corr || (!pairs && P > 0) || (!F_out && P > 0) || n <= 0 || P < 0) return GD_ERR_BADARG;
    for (int k = 0; k < 2 * P; ++k)
        if (pairs[k] < 0 || pairs[k] >= n) return GD_ERR_BADARG;
    gdb::PairScalars ps;
    gdb::pair_scalars(*settings, n, corr, pairs, P, ps);
    for (int k = 0; k < P; ++k) F_out[k] = ps.F[k];
    return GD_OK;
}

int gd_density2d_batch(gd_ctx* ctx, gd_ctx* twin, const gd_batch2d_settings* settings, gd_param2d* params, int32_t n,
                       const double* corr, const double* cov, const double* lag_probe, const int32_t* pairs, int32_t P,
                       gd_neff_exchange_fn exchange, void* exchange_user, void* grids_pinned, int64_t grids_doubles,
                       int32_t* status_pinned, double* meta, double* levels, int32_t* level_status, int32_t* tokens_out2) {
    GD_REQUIRE(ctx && settings && params && corr && cov && tokens_out2 && n > 0 && P >= 0, "bad argument");
    GD_REQUIRE(P == 0 || (pairs && grids_pinned && status_pinned && meta), "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(!twin || (twin != ctx && twin->cols == ctx->cols), "the second context must be attached to the first (gd_attach_samples)");
    GD_REQUIRE(ctx->w_sel == 0, "auxiliary weights are selected");
    GD_REQUIRE(!settings->want_levels || (settings->contours && settings->ncontours > 0 && levels && level_status), "contour levels requested without room for them");
    GD_HIP(hipSetDevice(ctx->device));
    int rc;
    try {
        gdb::Call call(state_of(ctx), kOps, ctx, twin, *settings, params, n, corr, cov, lag_probe, pairs, P, exchange, exchange_user,
                       (double*)grids_pinned, grids_doubles, status_pinned, meta, levels, level_status);
        rc = call.run(tokens_out2);
        if (rc) ctx->err = call.err;
    } catch (const std::exception& e) {
        rc = gd_fail(ctx, GD_ERR_NOMEM, "gd_density2d_batch: %s", e.what());
    }
    return rc;
}

int gd_density1d_batch(gd_ctx* ctx, const gd_density1d_settings* settings, gd_param2d* params, int32_t n, const int32_t* cols,
                       int32_t B, double* P_out, double* hist_out, double* meta) {
    GD_REQUIRE(ctx && settings && params && cols && P_out && meta && n > 0 && B > 0, "bad argument");
    GD_REQUIRE(ctx->cols, "no samples uploaded");
    GD_REQUIRE(ctx->w_sel == 0, "auxiliary weights are selected");
    GD_HIP(hipSetDevice(ctx->device));
    int rc;
    try {
        std::string err;
        rc = gdb::density1d_batch(state_of(ctx), kOps, kOps1D, ctx, *settings, params, n, cols, B, P_out, hist_out, meta, &err);
        if (rc) ctx->err = err;
    } catch (const std::exception& e) {
        rc = gd_fail(ctx, GD_ERR_NOMEM, "gd_density1d_batch: %s", e.what());
    }
    return rc;
}

int gd_batch2d_finish(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    if (!ctx->batch_state) return GD_OK;
    GD_HIP(hipSetDevice(ctx->device));
    return gdb::finish_all(*(gdb::State*)ctx->batch_state, kOps, ctx);
}

int gd_batch2d_exchanges(gd_ctx* ctx, int64_t* count_out) {
    GD_REQUIRE(ctx && count_out, "null argument");
    *count_out = 0;
    if (ctx->batch_state) {
        gdb::State& st = *(gdb::State*)ctx->batch_state;
        std::lock_guard<std::mutex> g(st.mu);
        *count_out = st.exchanges_entered;
    }
    return GD_OK;
}

int gd_batch2d_invalidate(gd_ctx* ctx) {
    GD_REQUIRE(ctx, "null context");
    if (!ctx->batch_state) return GD_OK;
    gdb::invalidate_index_columns(*(gdb::State*)ctx->batch_state);
    return GD_OK;
}

}  // extern "C"
