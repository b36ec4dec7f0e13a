/*
 * gdhip.h -- C ABI of libgdhip.so: the MI355X (gfx950) implementation of GetDist's weighted-sample
 * statistics + 1D/2D kernel-density hot path.
 *
 * The reference (GetDist 1.7.7) is pure Python and has NO FFI seam for this path: every
 * entry point below replaces a span of numpy/scipy calls inside the reference's Python methods; the
 * span is cited as file:line under /root/reference/getdist/.  The binding a maintainer would add is a
 * ctypes stub (getdist_amd/_lib.py, and INTEGRATION.md).
 *
 * Conventions: all entry points are extern "C", return 0 on success or a negative gd_status; no C++
 * exception crosses the ABI; sizes are int64_t; floating point is IEEE fp64 throughout; "d_" pointers
 * are DEVICE pointers obtained from gd_dev_alloc (opaque to the caller), everything else is host
 * memory owned by the caller.  A gd_ctx is bound to one device and one stream and is not thread-safe;
 * use one ctx per thread/GPU.  gd_last_error(ctx) returns a static/ctx-owned message for the last
 * failing call.
 */
#ifndef GDHIP_H
#define GDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gd_ctx gd_ctx;

typedef enum gd_status {
    GD_OK = 0,
    GD_ERR_BADARG = -1,    /* -> ValueError / SettingError in the Python layer */
    GD_ERR_NOMEM = -2,     /* -> MemoryError */
    GD_ERR_HIP = -3,       /* -> RuntimeError (message holds hipGetErrorString) */
    GD_ERR_EMPTY = -4,     /* "no samples in bin" -> DensitiesError (densities.py:83-84) */
    GD_ERR_SOLVER = -5,    /* bandwidth root-find failed -> fallback / BandwidthError (mcsamples.py:1258-1268) */
    GD_ERR_FFT = -6,       /* rocFFT failure */
    GD_ERR_NODEVICE = -7,  /* no HIP device visible */
    GD_ERR_TIMEOUT = -8    /* a collective / communicator set-up gave up waiting for a peer (the communicator was aborted and
                              dropped: renegotiate over the host application's own channel) -> parallel.CommTimeout */
} gd_status;

/* ---------------------------------------------------------------- context / memory ------------- */
int gd_device_count(void);
int gd_create(int device, gd_ctx** out);
void gd_destroy(gd_ctx* ctx);
const char* gd_last_error(gd_ctx* ctx);
const char* gd_version(void);
/* info[0]=CU count, [1]=LDS bytes/block, [2]=total HBM bytes, [3]=free HBM bytes, [4]=clock kHz, [5]=warp size */
int gd_device_info(gd_ctx* ctx, int64_t* info6);
int gd_sync(gd_ctx* ctx);
int gd_dev_alloc(gd_ctx* ctx, int64_t bytes, void** d_out);
int gd_dev_free(gd_ctx* ctx, void* d_ptr);
int gd_memcpy_h2d(gd_ctx* ctx, void* d_dst, const void* src, int64_t bytes);
int gd_memcpy_d2h(gd_ctx* ctx, void* dst, const void* d_src, int64_t bytes);
/* D2H on a second stream, ordered after the work queued so far on the compute stream; dst should be page-locked
 * (gd_host_alloc).  gd_copy_sync waits for all such copies. */
int gd_memcpy_d2h_async(gd_ctx* ctx, void* dst, const void* d_src, int64_t bytes);
int gd_copy_sync(gd_ctx* ctx);
/* gd_copy_mark: a point on the copy stream after every copy (and all compute-stream work) issued so far;
 * gd_copy_wait blocks until that point has been reached -- the caller's own result copies, not later ones (a batch
 * of results can be delivered while the next batch already computes).  16 marks are kept; an older token waits for
 * the newer mark that replaced it. */
int gd_copy_mark(gd_ctx* ctx, int32_t* token_out);
int gd_copy_wait(gd_ctx* ctx, int32_t token);
int gd_memcpy_d2d(gd_ctx* ctx, void* d_dst, const void* d_src, int64_t bytes);
int gd_memset(gd_ctx* ctx, void* d_dst, int value, int64_t bytes);
/* d_dst[k] = d_src[index[k]] for `count` items of item_bytes each (item_bytes % 16 == 0), one kernel launch;
 * stream-ordered (returns once enqueued; `index` may be released on return) */
int gd_gather_items(gd_ctx* ctx, void* d_dst, const void* d_src, const int32_t* index, int32_t count, int64_t item_bytes);
/* page-locked host memory for fast, asynchronous D2H of result grids.  Blocks of 32 MB and more are anonymous memory on
 * transparent huge pages (MADV_HUGEPAGE, the device's NUMA node preferred) registered with the runtime -- result copies into
 * 4-KB-page blocks allocated late in a process ran measurably slower --, smaller ones hipHostMalloc;
 * free with gd_host_free only. */
int gd_host_alloc(gd_ctx* ctx, int64_t bytes, void** out);
int gd_host_free(gd_ctx* ctx, void* ptr);
/* HIP-event timing on the ctx stream (bench.py measures kernels with these, not torch events) */
int gd_timer_start(gd_ctx* ctx);
int gd_timer_stop_ms(gd_ctx* ctx, double* ms_out);

/* ---------------------------------------------------------------- sample set -------------------
 * Replaces WeightedSamples.setSamples/_weightsChanged (chains.py:276-323): the library keeps its own
 * column-major (SoA) fp64 copy of the N x n sample array and of the weights in HBM.
 * X[i*row_stride + j*col_stride] is sample i of parameter j (strides in elements); weights may be
 * NULL (unit weights: chains.py:313-315).  Re-uploading replaces the previous set. */
int gd_upload(gd_ctx* ctx, const double* X, int64_t N, int64_t n, int64_t row_stride, int64_t col_stride,
              const double* weights);
int gd_num_rows(gd_ctx* ctx, int64_t* N, int64_t* n);
/* device pointer to column j (N doubles) / weights (NULL if unit) -- for tests and zero-copy users */
int gd_column_ptr(gd_ctx* ctx, int64_t j, void** d_out);

/* ---------------------------------------------------------------- weighted moments -------------
 * gd_weight_stats: norm=sum w (chains.py:312), max w (chains.py:1349), sum w^2 (chains.py:499,536),
 *   count of w > thresh (mcsamples.py:559-560).   out4 = {norm, max_w, sum_w2, n_above}
 * gd_col_stats: per column over rows [row_lo,row_hi): min, max (mcsamples.py:1434-1435), weighted mean
 *   (chains.py:379), weighted variance about that mean (chains.py:409-410). out = n x 4 {min,max,mean,var}
 * gd_cov: weighted means and covariance of the listed columns over rows [row_lo,row_hi)
 *   (chains.py:373-384 setMeans, :709-733 cov + mean_diffs :763-780); means_out m, cov_out m x m, norm_out 1; minmax_out
 *   (may be NULL) m x 2 = the columns' min / max over the range -- updateBaseStatistics then needs no separate
 *   gd_col_stats call (the variances are the diagonal).  Up to 207 columns in ONE read of the samples (round 6): deviations
 *   are taken from a provisional shift s (the mean of 256 strided rows), a column of ones rides in the slab, and
 *   mean = s + sum w (x - s) / sum w,  cov_ij = sum w (x_i - s_i)(x_j - s_j) / sum w - (mean_i - s_i)(mean_j - s_j):
 *   the reference's two passes (means, then deviations from the means) to rounding -- (mean - s)^2 is ~1e-3 of the
 *   variance, so the subtraction costs no digits.  208 columns and above: two passes as before. */
int gd_weight_stats(gd_ctx* ctx, int64_t row_lo, int64_t row_hi, double thresh, double* out4);
int gd_col_stats(gd_ctx* ctx, int64_t row_lo, int64_t row_hi, double* out);
int gd_cov(gd_ctx* ctx, const int32_t* cols, int32_t m, int64_t row_lo, int64_t row_hi, double* means_out,
           double* cov_out, double* norm_out, double* minmax_out);

/* ---------------------------------------------------------------- weighted quantiles -----------
 * Replaces initParamConfidenceData + confidence (chains.py:793-838: argsort, cumsum(w[idx]),
 * searchsorted(cumsum, target), x[idx[min(ix,N-1)]]) by a sort-free MSB radix select.
 * targets are cumulative-weight targets (norm*limfrac, computed by the caller exactly as the
 * reference does); out[c*k + t] = smallest sample value v of column cols[c] whose cumulative weight
 * sum_{x<=v} w reaches targets[c*k+t] (the maximum if none does). */
int gd_quantiles(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t row_lo, int64_t row_hi,
                 const double* targets, int32_t k, double* out);

/* gd_quantiles_mm: the same selection when the caller knows each column's minimum and maximum over (a superset of) the
 * row range -- minmax[2c], minmax[2c+1]; the base statistics have them (gd_cov's minmax_out).  Then ONE counting pass
 * over the monotone linear bucket index (int)((x - min) * nbuckets / (max - min)) in 32768 (unit weights) or 16384
 * (fp64 weights) LDS buckets narrows every target to a few hundred rows, which one collect pass gathers and a sorted walk
 * of the cumulative weight finishes exactly as above: two reads of the columns instead of four.  Taken for unit weights
 * and integer multiplicities (whose bucket sums are exact in any order of addition); minmax == NULL, real weights, a
 * degenerate range, more rows than the bucket lists can hold, or a list overflow (heavily tied data) take the radix
 * path of gd_quantiles; the result is the same sample value either way. */
int gd_quantiles_mm(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t row_lo, int64_t row_hi,
                    const double* targets, int32_t k, const double* minmax, double* out);

/* gd_quantiles_mm_probe: gd_quantiles_mm whose counting pass -- when it walks WHOLE columns of the sample set (row_lo = 0,
 * row_hi = N, the linear path) -- also serves two other consumers of the same read (round 6):
 *   - the first 8 autocovariance lag sums of getCorrelationLength's probe (chains.py:423-466, what gd_autocov_lags_batch(cols,
 *     probe_means, 0, 8) returns: sum_i d_i d_{i+l}, d = (x - mean) w) into probe_out[c * 8 + l]; *probe_done = 1 when they
 *     were computed (0: the select took another path -- call gd_autocov_lags_batch);
 *   - the columns' 16-bit bucket indices are kept on the device: the select's own collect pass, and later gd_prebin8_batch /
 *     gd_prebin8_hist2d / gd_prebin_batch of the same columns, read those 2 bytes per sample instead of the 8-byte value
 *     (a bucket that lies inside one bin maps through a table; the ~1 % that straddle a bin edge re-read the sample and
 *     take the exact fp64 route of mcsamples.py:1497 -- the indices are bit-equal either way).
 * probe_means / probe_out / probe_done may be NULL (then this is gd_quantiles_mm). */
int gd_quantiles_mm_probe(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t row_lo, int64_t row_hi,
                          const double* targets, int32_t k, const double* minmax, double* out, const double* probe_means,
                          double* probe_out, int32_t* probe_done);

/* ---------------------------------------------------------------- autocorrelation / N_eff ------
 * gd_autocov_lags: out[l] = sum_{i} d_i d_{i+k0+l}, d=(x-mean)*w, l<nlags -- the un-normalised lag
 *   sums convolve.autoConvolve (convolve.py:458-478) obtains by FFT for chains.py:441.
 * gd_kde_lag_sums: out[l] = sum_i exp(-(x_i-x_{i+k_l})^2 * inv4s2) w_i w_{i+k_l}
 *   (corr_k and the uncorrelated-term loop of getEffectiveSamplesGaussianKDE, chains.py:514-540). */
int gd_autocov_lags(gd_ctx* ctx, int32_t col, double mean, int64_t k0, int32_t nlags, double* out);
int gd_kde_lag_sums(gd_ctx* ctx, int32_t col, double inv4s2, const int64_t* lags, int32_t nlags, double* out);
/* 2D variant (getEffectiveSamplesGaussianKDE_2d, chains.py:576-635): out[l] = sum_i exp(-(d^T K d)/4) w_i w_{i+k_l},
 * d = (x_i - x_{i+k}, y_i - y_{i+k}), kinv3 = {K00, K01+K10, K11} with K = inv(cov)/h^2 */
int gd_kde_lag_sums_2d(gd_ctx* ctx, int32_t coli, int32_t colj, const double* kinv3, const int64_t* lags, int32_t nlags,
                       double* out);
/* batched over columns in one launch: out is ncols x nlags (per-column mean / inv4s2, shared lag list) */
int gd_autocov_lags_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* means, int64_t k0,
                          int32_t nlags, double* out);
int gd_kde_lag_sums_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* inv4s2, const int64_t* lags,
                          int32_t nlags, double* out);
/* the same lag sums restricted to rows [row_lo,row_hi) (one chain of a multi-chain set, mcsamples.py:941-962) */
int gd_autocov_lags_range_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* means, int64_t row_lo,
                                int64_t row_hi, int64_t k0, int32_t nlags, double* out);

/* ---------------------------------------------------------------- binning ----------------------
 * Index rule (mcsamples.py:1497): ix = (int)((x - binmin)/fine_width + 0.5), IEEE fp64, no FMA
 * contraction, true division.  Truncating rule (kde_bandwidth.py:86-87): (int)((x - range_min)/dx).
 * gd_hist1d: fused index + weighted bincount (mcsamples.py:1554) for ncols columns -> out ncols x F.
 * gd_bin_indices: writes the int32 indices (tests: bit-exact index parity; n_out_of_range counts
 *   indices outside [0,F)).
 * gd_prebin: u16 bin-index columns kept on the device for the batched 2D path; d_idx is N u16.
 * gd_hist2d: weighted 2D histogram hist[iy*F+ix] (mcsamples.py:1724-1728) for B pairs, direct from the
 *   fp64 columns (x index from colx with rule `round`, y from coly) -> d_hist B x F x F (device).
 * gd_hist2d_prebinned: same from u16 index columns produced by gd_prebin.  LDS counters: fp64 for real weights,
 *   u32 for unit or integral weights (exact), 16-bit packed for large unit-weight batches with exact overflow
 *   detection and a u32 redo of the affected pairs -- the result is always the exact weighted count.
 * gd_minmax_affine: min/max over samples of a*x_i + b*x_j (the sheared coordinate p2,
 *   mcsamples.py:1373 + kde_bandwidth.py:77-78); out 2*B.
 * gd_hist2d_sheared: rotated histogram of mcsamples.py:1372-1378: x index = trunc((x_i - xmin)/dx),
 *   y index = trunc(((r0*x_i + r1*x_j) - ymin)/dy). */
int gd_hist1d(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width,
              int32_t F, double* out);
/* the same with the histograms left in device memory (d_out: ncols x F doubles from gd_dev_alloc); returns when the
 * kernels are enqueued on the context's stream */
int gd_hist1d_dev(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width,
                  int32_t F, void* d_out);
int gd_bin_indices(gd_ctx* ctx, int32_t col, double binmin, double width, int32_t round_half, int32_t F,
                   int32_t* idx_out, int64_t* n_out_of_range);
int gd_prebin(gd_ctx* ctx, int32_t col, double binmin, double width, int32_t F, void* d_idx_u16);
/* Byte-index variant of the batched triangle binning (unit weights, fine_bins_2D = 256, the base grid of 98 % of a
 * triangle's pairs): gd_prebin8_batch writes (unsigned char)((x - binmin)/width + 0.5) for ncols columns in one launch
 * (d_idx_out[c]: device buffers of N bytes, 16-byte aligned) and reports in bad_out[c] how many samples fell outside
 * [0, F) -- zero by construction of the bin range (mcsamples.py:1486-1498), and required to be zero by
 * gd_hist2d_prebinned8, which builds B 256 x 256 histograms (device, B x 256 x 256 fp64, [y][x]) from pairs of those
 * columns: one block per pair, 16-bit counters packed two per LDS word, two bin addresses per v_perm_b32.  A wrapped
 * counter (more than 65535 samples in one bin) is detected exactly and reported as GD_ERR_SOLVER; the caller then uses
 * gd_prebin_batch + gd_hist2d_prebinned for that batch. */
int gd_prebin8_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                     void* const* d_idx_out, int64_t* bad_out);
int gd_hist2d_prebinned8(gd_ctx* ctx, int32_t B, const void* const* d_idx_x, const void* const* d_idx_y, void* d_hist);
/* gd_prebin8_hist2d: both of the above enqueued back to back with ONE wait: first the `ncols` byte index columns that have
 * to be (re)made (ncols may be 0), then the B histograms over index columns d_idx_x / d_idx_y (which may be among the
 * columns just made).  bad_out as gd_prebin8_batch; GD_ERR_SOLVER when any sample fell outside the grid or a counter
 * wrapped -- the histograms are then invalid and the caller takes gd_prebin_batch + gd_hist2d_prebinned. */
int gd_prebin8_hist2d(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width,
                      void* const* d_idx_out, int64_t* bad_out, int32_t B, const void* const* d_idx_x,
                      const void* const* d_idx_y, void* d_hist);

/* several index columns in one launch (d_idx_u16[c] receives column cols[c] binned with binmin[c], width[c]) */
int gd_prebin_batch(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* binmin, const double* width, int32_t F,
                    void* const* d_idx_u16);
int gd_hist2d(gd_ctx* ctx, int32_t B, const int32_t* colx, const int32_t* coly, const double* binminx,
              const double* widthx, const double* binminy, const double* widthy, int32_t F, void* d_hist);
int gd_hist2d_prebinned(gd_ctx* ctx, int32_t B, const void* const* d_idx_x, const void* const* d_idx_y, int32_t F,
                        void* d_hist);
int gd_minmax_affine(gd_ctx* ctx, int32_t B, const int32_t* coli, const int32_t* colj, const double* a,
                     const double* b, double* out);
int gd_hist2d_sheared(gd_ctx* ctx, int32_t B, const int32_t* coli, const int32_t* colj, const double* r0,
                      const double* r1, const double* xmin, const double* dx, const double* ymin, const double* dy,
                      int32_t F, void* d_hist);

/* gd_isj1d: kde_bandwidth.py:102-135 gaussian_kde_bandwidth_binned for B histograms (host, B x F) with effective
 *   sample numbers neff[b]: DCT-II of hist/sum, the Botev improved-Sheather-Jones fixed point (:59-73) solved by
 *   MINPACK hybrd for one unknown as scipy.optimize.fsolve(x0 = 0.53 N^-1/5, xtol = x0/20, factor = 1) runs it, then
 *   the brentq re-check on [0.019 N^-1/5, 0.5] when the root is below 0.019 N^-1/5 -- all inside one kernel.
 *   hfrac_out[b] = bandwidth in units of the bin range; status_out[b] = GD_OK, or GD_ERR_SOLVER where the
 *   reference returns None ("1D auto bandwidth failed": zero functional). */
int gd_isj1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, const double* neff, double* hfrac_out,
             int32_t* status_out);
/* the same reading the histograms in device memory (gd_hist1d_dev) */
int gd_isj1d_dev(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* neff, double* hfrac_out,
                 int32_t* status_out);

/* ---------------------------------------------------------------- 1D density -------------------
 * gd_dct1d: a = DCT-II(data/sum(data)) (scipy.fftpack.dct type 2, unnormalised), for the Botev ISJ
 *   fixed point (kde_bandwidth.py:113-117).  in/out host arrays B x F.
 * gd_density1d: everything after the bandwidth for B parameters (mcsamples.py:1588-1668): Gaussian
 *   taps (Kernel1D :129-135), 'same' convolution (convolve.py:196-202), boundary correction of order
 *   bco (0,1,2; mcsamples.py:1600-1647), mbc rounds of multiplicative bias correction (:1649-1666),
 *   normalize("max") (densities.py:71-92).  hist: B x F (host); smooth[b] in fine-bin units, winw[b];
 *   flags[b] bit0=has_limits_bot bit1=has_limits_top bit2=periodic.  P_out: B x F.
 *   status_out[b] = GD_OK or GD_ERR_EMPTY. */
int gd_dct1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, double* a_out);
int gd_density1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, const double* smooth, const int32_t* winw,
                 const int32_t* flags, int32_t bco, int32_t mbc, double* P_out, int32_t* status_out);
/* the same reading the histograms in device memory (gd_hist1d_dev); P_out is on the host */
int gd_density1d_dev(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* smooth, const int32_t* winw,
                     const int32_t* flags, int32_t bco, int32_t mbc, double* P_out, int32_t* status_out);

/* ---------------------------------------------------------------- 2D bandwidth -----------------
 * gd_kopt2d: KernelOptimizer2D.__init__ + the psi functionals get_h needs (kde_bandwidth.py:146-270)
 *   for B square histograms (device, B x F x F): a2 = dct2d(data/sum)[1:,1:]^2 (:151), t* by Brent's
 *   method on the 2D fixed point over [0,0.1] with xtol=1e-6 (:162,177-196; scipy brentq semantics:
 *   rtol=4eps, maxiter=100), the fallback_t rules (:164-175; fallback_t<=0 means None),
 *   psi_02, psi_20, psi_11 = func2d at t* (:245-247), and when do_corr[b]: psi_00 (:267) and the odd
 *   functionals psi_13, psi_31 from |fft2|^2 (:156-157,198-214,269-270).
 *   Then KernelOptimizer2D.get_h (:234-306) on the device, one wavefront per pair: the closed-form bandwidths
 *   (:245-252) and, when do_corr[b], the two TNC minimisations of the AMISE (:276-302) started from corr[b]
 *   (scipy.optimize.minimize(method="TNC") ported evaluation-faithfully, csrc/solvers.hpp), with the reference's
 *   acceptance rules.
 *   out: B x 12 = {t_star, p02, p20, p11, p00, p13, p31, status(0 ok, <0 gd_status), hx, hy, corr,
 *   get_h status (0 ok, GD_ERR_BADARG = "bias not positive definite", GD_ERR_SOLVER = no functionals)};
 *   hx, hy are fractions of the histogram's bin range. */
int gd_kopt2d(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr,
              const double* fallback_t, const double* corr, double* out);
/* The same work in two stream-ordered stages, for callers that keep several launches in flight:
 *   gd_kopt2d_enqueue: everything up to the functionals (sums, DCT, power spectra, the fixed point) on ctx's stream; returns
 *     at once.  d_rows: device block of B x GD_KOPT_BLOCK_DOUBLES doubles: the B x 12 result rows
 *     {t_star, p02, p20, p11, p00, p13, p31, status, ...} followed by get_h's per-pair inputs, which this call uploads.
 *     *ticket_out identifies the point on ctx's stream behind those kernels (32 tickets are kept).
 *   gd_kopt2d_finish: get_h (the closed forms and the TNC minimisations) for those rows on `ctx`'s stream -- which may be
 *     ANOTHER context of the same device than stage_a_ctx: its stream waits for the ticket, so the serial TNC stage of one
 *     launch runs beside the next launch's DCT / fixed point -- and the finished B x 12 rows in `out` (host).  Blocks
 *     until they are there. */
#define GD_KOPT_BLOCK_DOUBLES 15
int gd_kopt2d_enqueue(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* neff, const int32_t* do_corr,
                      const double* fallback_t, const double* corr, void* d_rows, int32_t* ticket_out);
int gd_kopt2d_finish(gd_ctx* ctx, gd_ctx* stage_a_ctx, int32_t ticket, int32_t B, void* d_rows, double* out);
/* gd_get_h: the get_h stage alone, from host arrays: psi is B x 6 = {p02, p20, p11, p00, p13, p31};
 *   out: B x 4 = {hx, hy, corr, status}. */
int gd_get_h(gd_ctx* ctx, int32_t B, const double* psi, const double* neff, const double* corr, const int32_t* do_corr,
             double* out);

/* ---------------------------------------------------------------- 2D density -------------------
 * gd_density2d: everything after the bandwidth for B pairs sharing F (mcsamples.py:1857-1990):
 *   window synthesis from (rx, ry, corr, winw) (:1863-1867), zero-padded linear FFT convolution via
 *   rocFFT (convolve.py:405-436), linear boundary correction of order bco (0/1) where a parameter has a
 *   limit (:1905-1961), mbc rounds of multiplicative bias correction (:1963-1976), normalize("max") (:1990).
 *   flags[b]: bit0/1 = x has_limits_bot/top, bit2/3 = y bot/top (non-periodic axes only, :1688-1703);
 *   bit4/5 = x/y periodic (circular convolution on the folded grid, convolve.py:215-323; must be equal for the
 *   whole batch); bit6 = has_prior, i.e. the boundary-correction block applies (defaults to "any of bits 0-3").
 *   d_hist: B x F x F (device, [y][x]); d_P_out: B x F x F (device).  status_out[b] as gd_density1d. */
int gd_density2d(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* rx, const double* ry,
                 const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc,
                 void* d_P_out, int32_t* status_out);
/* gd_density2d_enqueue: the same work, but the call returns as soon as it is enqueued on the context's stream, so
 *   the host prepares the next batch (and builds result objects) while this one computes.  The per-pair table is
 *   staged through page-locked memory owned by the context; `status_pinned` must be page-locked (gd_host_alloc) and
 *   is valid after gd_sync, or after gd_copy_sync of a gd_memcpy_d2h_async issued after this call.  d_hist and
 *   d_P_out must stay allocated until then. */
int gd_density2d_enqueue(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const double* rx, const double* ry,
                         const double* corr, const int32_t* winw, const int32_t* flags, int32_t bco, int32_t mbc,
                         void* d_P_out, int32_t* status_pinned);

/* gd_density2d_enqueue_indexed: the same for a batch that is NOT contiguous in d_hist: pair b convolves the histogram
 *   hist_index[b] of the block (B_src x F x F) -- the pairs of one frame-size class picked out of a grid-size class, without
 *   copying their histograms together first. */
int gd_density2d_enqueue_indexed(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const int32_t* hist_index,
                                 const double* rx, const double* ry, const double* corr, const int32_t* winw,
                                 const int32_t* flags, int32_t bco, int32_t mbc, void* d_P_out, int32_t* status_pinned);

/* ---------------------------------------------------------------- second lane -----------------
 * A context is one stream; a second context on the same device gives a second, concurrent lane of work over the
 * SAME resident sample set (independent pairs of a triangle are dealt to two lanes so that one lane's host-side
 * scalar work and result copies hide behind the other lane's kernels).
 * gd_attach_samples: ctx borrows owner's device columns / weights (no copy); owner must outlive ctx or be
 *   re-uploaded only after ctx is destroyed or re-attached.  The extra columns (gd_set_extra_column) are shared too.
 * gd_bind_thread: make ctx's device current for the calling host thread (HIP's current device is per thread);
 *   call once from every additional thread that uses ctx. */
int gd_attach_samples(gd_ctx* ctx, gd_ctx* owner);
int gd_bind_thread(gd_ctx* ctx);

/* ---------------------------------------------------------------- contour levels ---------------
 * gd_contour_levels: densities.py:19-56 getContourLevels(P, contours, half_edge=True) for B grids
 *   (device, B x F x F): out[b*nc + c] = the density level enclosing contours[c] of the half-edge-weighted
 *   mass (argsort of the un-halved grid, cumulative half-weighted mass, linear interpolation between the two
 *   bracketing sorted entries).  status_out[b]: GD_OK; GD_ERR_EMPTY = "Contour level outside plotted ranges"
 *   (:51-52); GD_ERR_SOLVER = more than 1024 exactly equal values at the level (do that grid on the host). */
int gd_contour_levels(gd_ctx* ctx, int32_t B, int32_t F, const void* d_P, const double* contours, int32_t nc, double* out,
                      int32_t* status_out);

/* ---------------------------------------------------------------- credible limits 1D -----------
 * gd_limits1d: densities.py:186-248 Density1D.initLimitGrids + getLimits for B densities on regular grids
 *   (host arrays: P is B x F, x0[b] the first grid abscissa, spacing[b] the grid step).  Each density is refined
 *   `factor` times (<= 0: the reference default max(2, 20000 / F)) with the not-a-knot cubic spline through its F
 *   values; per contour c the density level below which (1 - contours[c]) of the refined mass lies is read off the
 *   ranked refined values (interpolating towards the NEXT ranked value, :227), and the outermost crossings of that
 *   level are located.  out[(b*nc + c)*4 ..] = {lower, upper, has_min, has_top} with has_* = 1.0 where the density at
 *   that end of the grid is still >= the level (no crossing: the limit is the grid end).  status_out[b]: GD_OK or
 *   GD_ERR_SOLVER (level beyond the ranked values / no crossing found, where the reference raises IndexError). */
int gd_limits1d(gd_ctx* ctx, int32_t B, int32_t F, const double* P, const double* x0, const double* spacing,
                const double* contours, int32_t nc, int32_t factor, double* out, int32_t* status_out);

/* ---------------------------------------------------------------- auxiliary vectors ------------
 * The reference lets most statistics take an arbitrary vector instead of a column index (`_makeParamvec`,
 * chains.py:325-337), a row filter `where=` (chains.py:666-780) or alternative weights (chains.py:793-838).
 * gd_set_extra_column: copy a host vector (N rows) into spare column `slot` (0..3); it is then addressed as
 *   column index  n + slot  by every entry point that takes column indices.
 * gd_aux_weights: copy a host weight vector (N rows, e.g. weights*where) into the auxiliary weight buffer
 *   (shared with gd_like_weights); gd_select_weights(ctx, 1) makes every weighted entry point use it.
 * gd_col_minmax: out[2c], out[2c+1] = min, max of column cols[c] over the rows of [lo,hi) whose value in
 *   column cond_col is < cond_below (cond_col < 0: all rows; +inf / -inf when no row qualifies) -- the
 *   N-dimensional confidence-region limits of _setLikeStats (mcsamples.py:2263-2274) with cond_col = the
 *   loglikes column and cond_below = its weighted quantile from gd_quantiles. */
int gd_set_extra_column(gd_ctx* ctx, int32_t slot, const double* x);
int gd_aux_weights(gd_ctx* ctx, const double* w);
int gd_col_minmax(gd_ctx* ctx, const int32_t* cols, int32_t ncols, int64_t lo, int64_t hi, int32_t cond_col,
                  double cond_below, double* out);

/* ---------------------------------------------------------------- thinned chains ---------------
 * Raftery-Lewis and CorrSteps (mcsamples.py:1039-1221) work on weight-one thinnings of integer-weight chains
 * (chains.py:878-916 thin_indices_single_samples).
 * gd_weights_integral: *out = 1 when every sample weight is a non-negative integer (or weights are absent).
 * gd_thin_rows: the thinned row list of chain rows [lo,hi) for `factor`, as int32 global row indices into
 *   d_rows (device, `capacity` entries).  unique_mode = 1 is the reference's branch for factor >= max weight
 *   (np.unique(cumsum // factor) first indices, :889-892), 0 its sequential loop (:894-914: row i once per
 *   multiple of factor in (C_{i-1}, C_i]).  Both come from one cached prefix sum of the weights.
 * gd_binary_transitions: for each column c and threshold t (thresholds: ncols x nthr, nthr <= 4), with
 *   b[k] = (x[rows[k]] >= u ? 0 : 1): counts_out[(c*nthr+t)*12 + 0..7] = np.bincount(4 b[k-2] + 2 b[k-1] + b[k])
 *   (:1065-1066) and [8..11] = np.bincount(2 b[k-1] + b[k]) (:1120-1122).
 * gd_thinned_lag_sums: out[c*maxoff + off-1] = sum_k (x[rows[k+off]]-means[c]) (x[rows[k]]-means[c])  (:1204-1206). */
int gd_weights_integral(gd_ctx* ctx, int32_t* out);
int gd_thin_rows(gd_ctx* ctx, int64_t lo, int64_t hi, int64_t factor, int32_t unique_mode, void* d_rows, int64_t capacity,
                 int64_t* count_out);
int gd_binary_transitions(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const void* d_rows, int64_t K,
                          const double* thresholds, int32_t nthr, int64_t* counts_out);
int gd_thinned_lag_sums(gd_ctx* ctx, const int32_t* cols, int32_t ncols, const double* means, const void* d_rows, int64_t K,
                        int32_t maxoff, double* out);

/* ---------------------------------------------------------------- mean likelihoods -------------
 * The optional `meanlikes` branches of get1DDensityGridData / get2DDensityGridData.
 * gd_like_weights: build the device vector  weights*exp(mean_loglike - loglikes)  (mode 0; mcsamples.py:1560,
 *   1830)  or  weights*loglikes  (mode 1, shade_likes_is_mean_loglikes; :1558) from the host column
 *   loglikes (N rows); *sum_out (optional) = the sum of the vector, so that mode 1 with sum_out gives
 *   mean_loglike = sum/norm (chains.py:380-383).  loglikes == NULL drops the vector.
 * gd_select_weights: which = 1 makes every histogram entry point (gd_hist1d, gd_hist2d, gd_hist2d_prebinned)
 *   accumulate the like weights instead of the sample weights (the `np.bincount(..., weights=w)` of
 *   :1561,1831); which = 0 restores the sample weights.  Must be 0 for everything else.
 * gd_likes1d: mcsamples.py:1672-1682 for B parameters: rawbins = conv(hist, Win); likehist/P where P>0,
 *   smoothed, times P/rawbins; the exp(-(l - min l)) transform when shade_mean_loglikes; / max.
 *   hist, likehist, P (the finished density of gd_density1d), likes_out: host B x F; smooth/winw/flags as
 *   gd_density1d.
 * gd_likes2d: mcsamples.py:1886-1903,2004-2006 for B pairs: bin2Dlikes = conv(likehist, Win) [with the
 *   mbc re-smoothing of :1890-1897 when mbc != 0], divided by the uncorrected conv(hist, Win) where that
 *   exceeds 1e-4 of its maximum (0 elsewhere), / max.  d_hist, d_likehist, d_likes_out: device B x F x F;
 *   rx/ry/corr/winw/flags as gd_density2d (only the periodic bits matter).  The convolutions are evaluated
 *   by direct summation, not FFTs: the same linear maps, but free of the cancellation noise whose sign decides
 *   the reference's `bin2Dlikes > 0` mask at low-likelihood pixels ("mean likelihoods"). */
int gd_like_weights(gd_ctx* ctx, const double* loglikes, int32_t mode, double mean_loglike, double* sum_out);
int gd_select_weights(gd_ctx* ctx, int32_t which);
int gd_likes1d(gd_ctx* ctx, int32_t B, int32_t F, const double* hist, const double* likehist, const double* P,
               const double* smooth, const int32_t* winw, const int32_t* flags, int32_t shade_mean_loglikes,
               double* likes_out, int32_t* status_out);
int gd_likes2d(gd_ctx* ctx, int32_t B, int32_t F, const void* d_hist, const void* d_likehist, const double* rx,
               const double* ry, const double* corr, const int32_t* winw, const int32_t* flags, int32_t mbc,
               void* d_likes_out, int32_t* status_out);

/* gd_density2d_masked: gd_density2d for ONE pair whose prior mask was edited by a user callback
 *   (mask_function, mcsamples.py:1767-1770,1907-1919): mask_bc = the (F+2 winw)^2 mask after _setEdgeMask2D
 *   (host, row-major; NULL when no boundary correction), mask_mbc = the same after _setAllEdgeMask2D (NULL when
 *   mbc == 0), zero_mask = F^2 bytes, 1 where mask < 1e-8 (bool_mask: the bias correction does not divide there and
 *   the density is zeroed at the end, :1973-1979).  The mask moments are summed directly.  Non-periodic axes only. */
int gd_density2d_masked(gd_ctx* ctx, int32_t F, const void* d_hist, double rx, double ry, double corr, int32_t winw,
                        int32_t flags, int32_t bco, int32_t mbc, const double* mask_bc, const double* mask_mbc,
                        const unsigned char* zero_mask, void* d_P_out, int32_t* status_out);

/* ---------------------------------------------------------------- stand-alone convolutions -------
 * The device-backed forms of getdist/convolve.py's public functions (host arrays in and out; the Python module
 * getdist_amd/convolve.py does the padding, the centring roll and the mode slices exactly as convolve.py:196-444).
 * gd_circ_convolve: out = irfft(rfft(a) * rfft(b)) for two real frames of n0 x n1 doubles (n0 == 1: one dimension,
 *   any n1 >= 2) -- the transform pair behind convolveFFT / convolveFFTn / convolve1D_periodic / convolve2D_periodic
 *   (convolve.py:371-436, 215-367), through rocFFT.
 * gd_convolve1d_direct: out_full[i] = sum_j x[j] y[i-j], i < nx + ny - 1: what np.convolve evaluates when an operand
 *   is shorter than 1000 samples (convolve.py:199-202).
 * gd_autoconvolve: autoConvolve(x, n, normalize) (convolve.py:458-478): result[k] = sum_i x_i x_{i+k} [/ (N - k)],
 *   k < n, through a real FFT of length s = nearestFFTnumber(2 N) (passed in: the table lives in the Python module).
 *   x_host != NULL: that vector of nx values; else the resident column `col` as (x - mean) * w (use_weights != 0) --
 *   the vector getAutocorrelation builds (chains.py:439-441) -- without it ever existing on the host. */
int gd_circ_convolve(gd_ctx* ctx, int32_t n0, int32_t n1, const double* a, const double* b, double* out);
int gd_convolve1d_direct(gd_ctx* ctx, const double* x, int64_t nx, const double* y, int64_t ny, double* out_full);
int gd_autoconvolve(gd_ctx* ctx, int32_t col, double mean, int32_t use_weights, const double* x_host, int64_t nx, int64_t s,
                    int64_t n, int32_t normalize, double* out);

/* gd_like_stats: the sample-likelihood numbers of MCSamples._setLikeStats (mcsamples.py:2216-2243) from the column
 *   `col` holding loglikes (an extra column, gd_set_extra_column) and the selected weights, in two passes:
 *   out8 = {min L, max L, sum w, sum w L, sum w L^2, sum w exp(L - min L), sum w exp(-(L - min L)),
 *           first row index attaining min L (np.argmin)}. */
int gd_like_stats(gd_ctx* ctx, int32_t col, double* out8);


/* ---------------------------------------------------------------- one native entry for a batch of pairs ----------
 * gd_density2d_batch: MCSamples.get2DDensityGridData (mcsamples.py:1748-2010) for P parameter pairs in ONE call, with
 * every host decision of the reference made inside the library (csrc/batch2d.hpp):
 *   - the per-pair correlation handling, angle_scale and the up-scaling of the fine grid (mcsamples.py:1796-1816),
 *   - the bin edges of _binSamples (mcsamples.py:1486-1498) and the 2D histograms (:1724-1728), index columns cached
 *     per (parameter, grid size) and shared by the pairs of a triangle,
 *   - the effective sample numbers of the parameters that have none yet (_get1DNeff -> getEffectiveSamplesGaussianKDE,
 *     chains.py:477-574, mcsamples.py:1230-1235), on the first stream BESIDE the binning on the second,
 *   - getAutoBandwidth2D (mcsamples.py:1285-1419): branch A (sheared re-binning + Cholesky de-rotation), B (rule of
 *     thumb), C (KernelOptimizer2D on the pair's own grid), the fallback widths and the higher-order widening,
 *   - window half-widths and frame sizes, batches of equal frame size, the convolution / boundary correction /
 *     multiplicative bias correction / normalisation of gd_density2d_enqueue (mcsamples.py:1857-1990) on two streams,
 *     the optimiser's launch cut in two so that the first half is convolved while the second is optimised,
 *   - the D2H copies of the finished grids into ONE page-locked block of the caller, batch by batch on the copy streams.
 * The call returns when everything is enqueued; the grids are complete after gd_copy_wait(ctx, tokens_out[0]) and, when
 * `twin` was given, gd_copy_wait(twin, tokens_out[1]).
 *
 * ctx    : the context holding the samples.  twin: a second context on the same device attached to it
 *          (gd_attach_samples) = the second stream; NULL runs everything on one stream.
 * params : n records, index = column number; only the columns that occur in `pairs` are read.  `neff` is in/out: NaN on
 *          entry = not known yet, computed here when the bandwidth is automatic (smooth_scale_2D < 0).  `owned` != 0
 *          marks the parameters whose N_eff THIS rank computes in a multi-rank job; the others arrive through `exchange`.
 * corr / cov : the n x n correlation and covariance matrices of the columns (chains.py:155-169, 709-733), row-major.
 * lag_probe  : optional n x 8 autocovariance lag sums (gd_autocov_lags_batch of all columns, lags 0..7) a caller may have
 *          prefetched beside its quantile select; NULL = computed here.
 * pairs  : P x 2 column indices (x, y).
 * exchange : optional; called exactly once per call, on the calling thread, after this rank's own N_eff kernels are
 *          through: it receives the n-vector of N_eff values (NaN = unknown here) and fills in the other ranks' values
 *          (an all-gather); return non-zero to abort.  NULL with settings->comm_exchange and a communicator on the context
 *          (gd_comm_init): the library exchanges them itself (ncclAllReduce on the context's stream).
 * grids_pinned : page-locked block of >= sum F_k^2 doubles (F_k from gd_batch2d_grid_sizes); pair k's grid lands at
 *          meta[k][1] doubles from its start, F x F row-major, [y][x].
 * status_pinned : P int32, page-locked; after the copies have landed status_pinned[(int)meta[k][30]] is GD_OK or
 *          GD_ERR_EMPTY ("no samples in bin", densities.py:83-84) for pair k.  Until a batch of convolutions has run its
 *          words hold 0x7fffffff: the call pre-sets them and the batch's last launch overwrites them -- the library's own
 *          threads read the words of the first batches to learn that those batches have RUN (the deferred shear chain starts
 *          then), so the block has to be host-visible while the call runs, as page-locked memory is.
 * meta   : P x GD_BATCH2D_META doubles:
 *          [0] F  [1] offset of the grid  [2..4] (hx, hy, corr) of the kernel in parameter units (NaN when the scale is fixed)
 *          [5] bandwidth branch 0/1/2 = A/B/C (-1: fixed scale)  [6..17] the optimiser's record as gd_kopt2d writes it
 *          [18..20] window scales rx, ry in fine-bin units and the window's correlation  [21] window half-width
 *          [22] bits: 1 = "Parameters are 100% correlated", 2 = "fine_bins_2D not large enough", 4 = fallback widths used
 *          [23..26] xbinmin, xbinmax, ybinmin, ybinmax  [27] the pair's correlation as used for the grid  [28] nbin2D
 *          [29] 0 / 1 = the grid's copy runs on ctx's / twin's copy stream  [30] index into status_pinned  [31] N_eff used
 * levels / level_status : when settings->want_levels, P x ncontours contour levels (gd_contour_levels) and P states.
 * Errors: GD_ERR_BADARG "bias not positive definite" (kde_bandwidth.py:229-230) and setting errors; GD_ERR_SOLVER when
 *   raise_on_bandwidth_errors and the optimiser found no root; GD_ERR_FFT + 1000 * ... never; a parameter whose chain
 *   correlation outlasts the 8-lag probe returns GD_BATCH2D_NEED_NEFF with its column in gd_last_error -- the caller
 *   computes that N_eff (getCorrelationLength's long route) and calls again. */
#define GD_BATCH2D_META 32
#define GD_BATCH2D_NEED_NEFF (-20)

typedef struct gd_param2d {
    double range_min, range_max; /* par.range_min / range_max after _initParam (mcsamples.py:1421-1484) */
    double param_min, param_max; /* column extrema */
    double sigma_range, err;     /* par.sigma_range, par.err */
    double mean, var;            /* weighted mean and variance of the column (the N_eff probe) */
    double neff;                 /* par.N_eff_kde; NaN = unknown (in/out) */
    int32_t has_limits_bot, has_limits_top, periodic;
    int32_t owned; /* bit 0: this rank computes the parameter's N_eff; bit 1: no bandwidth warning / error for it (1D) */
} gd_param2d;

typedef struct gd_batch2d_settings {
    int32_t fine_bins_2D, boundary_correction_order, mult_bias_correction_order, num_bins_2D;
    double smooth_scale_2D, max_corr_2D;
    double norm, sum_w2;                /* sum w, sum w^2 of the sample weights (chains.py:312,499) */
    int32_t uncorrelated_sampler;       /* sampler in ("nested", "uncorrelated"): N_eff = norm^2 / sum_w2 (chains.py:507-508) */
    int32_t raise_on_bandwidth_errors;
    int32_t want_levels, ncontours;     /* contour levels on the device instead of lazily delivered grids */
    const double* contours;
    /* choreography (defaults in parentheses; 0 = default) */
    int32_t two_streams_min;            /* (64)  pairs from which the convolution uses both streams */
    int32_t two_streams_split;          /* (400) pairs above which the optimiser's launch is cut in two */
    int32_t kopt_split_min;             /* (256) ... when the base grid's launch has at least this many pairs */
    double kopt_first_fraction;         /* (0.5) */
    int32_t first_batch, max_batch;     /* (128, 320) grids in the first / in every further convolution batch */
    double max_batch_bytes;             /* (24e9) device scratch a batch may take */
    int32_t comm_exchange;              /* 1: exchange the N_eff values over the context's communicator (gd_comm_init): one
                                           sum all-reduce of n doubles, every rank contributing the parameters it owns */
    int32_t bandwidths_only;            /* 1: stop when the bandwidths are known -- meta[2..5], [18..22], [31] are filled, nothing is
                                           convolved or copied (a caller that edits each pair's prior mask, mcsamples.py:1767-1770,
                                           needs the window half-widths first and then convolves through gd_density2d_masked) */
    /* optional per-pair inputs (NULL = none); with smooth_scale_2D < 0 only */
    const double* pair_neff;            /* P effective sample numbers instead of min(N_eff x, N_eff y): the caller's 2D estimate
                                           under use_effective_samples_2D (mcsamples.py:1322-1328, chains.py:576-635) */
    const double* bandwidths;           /* P x 3 (hx, hy, corr) in parameter units instead of getAutoBandwidth2D */
    int32_t results_in_flight;          /* 1: the caller has not waited for the PREVIOUS batched call's result copies (a stream of
                                           calls): this call's first grids queue behind those copies anyway, so the deferred
                                           shear chain starts when the first part's convolution is enqueued (throughput) instead
                                           of when it has run (0: the first grids of THIS call are what its delivery waits for) */
} gd_batch2d_settings;

typedef int (*gd_neff_exchange_fn)(void* user, double* neff_n, int32_t n);

/* F_out[k] = the fine grid size pair k will get (mcsamples.py:1796-1816: 256, or 192 * (3 / angle_scale) // 3 for
 * strongly correlated pairs); host only. */
int gd_batch2d_grid_sizes(const gd_batch2d_settings* settings, int32_t n, const double* corr, const int32_t* pairs,
                          int32_t P, int32_t* F_out);
int gd_density2d_batch(gd_ctx* ctx, gd_ctx* twin, const gd_batch2d_settings* settings, gd_param2d* params, int32_t n,
                       const double* corr, const double* cov, const double* lag_probe, const int32_t* pairs, int32_t P,
                       gd_neff_exchange_fn exchange, void* exchange_user, void* grids_pinned, int64_t grids_doubles,
                       int32_t* status_pinned, double* meta, double* levels, int32_t* level_status, int32_t* tokens_out2);
/* ---------------------------------------------------------------- one native entry for a batch of 1D densities ----
 * gd_density1d_batch: MCSamples.get1DDensityGridData (mcsamples.py:1500-1686) for B parameters in ONE call, the host
 * decisions of the reference made inside the library (csrc/batch1d.hpp):
 *   - the bin edges of _binSamples (mcsamples.py:1486-1496) and the fused index + weighted bincount (:1554),
 *   - with smooth_scale_1D <= 0: the effective sample numbers of the parameters that have none yet (as gd_density2d_batch
 *     computes them), the Botev ISJ bandwidth (gd_isj1d) and the scalar tail of getAutoBandwidth1D (:1256-1283: the rule
 *     of thumb when the solver fails or the width is very small, the higher-order rescaling); otherwise the fixed scales
 *     of :1572-1582,
 *   - the smoothing scale in fine-bin units, the window half-width, and everything of gd_density1d (:1588-1668).
 * The histograms stay in device memory between the three stages.  Blocking: P_out is complete on return.
 *
 * params : n records as for gd_density2d_batch, index = column number; only the listed columns are read.  `neff` is in/out
 *          (NaN = not known; computed here when the bandwidth is automatic; GD_BATCH2D_NEED_NEFF as for the 2D entry).
 * cols   : B column indices (a parameter may be listed once).  P_out : B x F doubles (host), normalised to max = 1.
 * hist_out : optional B x F doubles (host): the histograms (the mean-likelihood profiles of gd_likes1d need them).
 * meta   : B x GD_BATCH1D_META doubles: [0] binmin [1] binmax [2] the bandwidth h in units of the bin range (par.kde_h;
 *          NaN when the scale is fixed) [3] smooth_1D in fine-bin units [4] window half-width
 *          [5] bits: 1 = "1D auto bandwidth failed" (solver returned None), 2 = very small or failed: rule-of-thumb
 *          fallback used, 4 = "fine_bins not large enough to well sample smoothing scale"
 *          [6] N_eff used (NaN when the scale is fixed) [7] GD_OK or GD_ERR_EMPTY ("no samples in bin") for the density
 *          [8] with bit 2: the solver's own width that the fallback replaced (NaN = it returned None) -- the "h=" of the
 *          reference's message "auto bandwidth for X very small or failed (h=..,N_eff=..). Using fallback (h=..)" (:1262).
 *          A parameter whose record has (owned & 2) set is one the caller lists in no_warning_params (or a chi2 parameter
 *          under no_warning_chi2_params, :1259-1261): the fallback is then taken without bit 2 and without an error.
 * Errors: GD_ERR_BADARG for setting errors ("Parameter range is <= 0", an unknown boundary_correction_order); GD_ERR_SOLVER
 *   when raise_on_bandwidth_errors is set and the fallback would have been used. */
#define GD_BATCH1D_META 9

typedef struct gd_density1d_settings {
    int32_t fine_bins, num_bins, boundary_correction_order, mult_bias_correction_order;
    double smooth_scale_1D;
    double norm, sum_w2;           /* sum w, sum w^2 of the sample weights (N_eff of parameters that have none yet) */
    int32_t uncorrelated_sampler;  /* as gd_batch2d_settings */
    int32_t raise_on_bandwidth_errors;
} gd_density1d_settings;

int gd_density1d_batch(gd_ctx* ctx, const gd_density1d_settings* settings, gd_param2d* params, int32_t n, const int32_t* cols,
                       int32_t B, double* P_out, double* hist_out, double* meta);

/* Completes the last batched call(s) of the context (waits for their result copies, hands their device blocks back to
 * the library's pool); gd_batch2d_invalidate additionally marks every cached index column stale (the next call bins
 * again -- what a benchmark does between steps, and what gd_upload does by itself). */
int gd_batch2d_finish(gd_ctx* ctx);
int gd_batch2d_invalidate(gd_ctx* ctx);
/* How many times gd_density2d_batch has ENTERED its own N_eff collective (settings->comm_exchange) on this context, counted
 * when the all-reduce is issued -- i.e. also for a call that failed afterwards (a matrix that is not positive definite, a
 * bandwidth error, GD_BATCH2D_NEED_NEFF from the columns nobody owned).  A multi-rank caller reads it around a call to
 * learn whether this rank has already taken part in the step's collective, so that its error path neither skips nor
 * repeats it ("exactly once per rank and step"; the reference has no such exchange: mcsamples.py:1230-1235 is per process). */
int gd_batch2d_exchanges(gd_ctx* ctx, int64_t* count_out);


/* ---------------------------------------------------------------- multi-GPU: RCCL over xGMI ------------------------
 * One process per GPU.  The path has no data-path collective: pairs are partitioned and every rank copies
 * its own grids to its host.  What is exchanged are small per-rank vectors -- the partial moments of a row share
 * (chains.py:709-733 pooled over ranks), the per-parameter state _initParam leaves (mcsamples.py:1421-1484; ~12 doubles
 * per parameter) and the N_eff values (one double per parameter) -- by ALL-GATHER, and additive partial tables
 * (histograms, bucket counts: chains.py:793-838, mcsamples.py:1486-1498,1724-1728) by SUM ALL-REDUCE.
 * gd_comm_unique_id: rank 0 obtains the 128-byte RCCL id and hands it to the other ranks by whatever means the host
 *   application has (torch.distributed's store, MPI, a file); gd_comm_init: every rank then joins with it (collective).
 * gd_comm_allgather: recv[r * count ..] = rank r's `send` (host vectors; staged through the context's device block, the
 *   collective itself runs on device memory on the context's stream: ncclAllGather).  gd_comm_allreduce_sum likewise.
 * The *_dev forms take device pointers and return once enqueued on the context's stream.
 * With a communicator on the context gd_density2d_batch exchanges the N_eff values itself when `exchange` is NULL. */
/* gd_comm_rccl_path: the file the collectives resolve to and whether it was already mapped by the process (the host
 *   application's RCCL, e.g. PyTorch's bundled copy: the library never maps a second RCCL beside it).
 * gd_comm_init runs ncclCommInitRank under a watchdog and the host-vector collectives poll the stream and the
 *   communicator's asynchronous error state instead of blocking: a rank that never arrives or dies is an error return
 *   (GD_ERR_HIP, after GDHIP_COMM_TIMEOUT_S seconds, default 120; the communicator is aborted and dropped), not a hang.
 * gd_upload_shard + gd_comm_share_columns: sample distribution over xGMI -- rank r uploads only its block of columns
 *   [first[r], first[r+1]) (column-major, column stride col_stride >= N; weights by every rank), then every rank
 *   broadcasts its block to the others (W ncclBroadcast in one group).  After gd_comm_share_columns the context holds the
 *   full set exactly as after gd_upload of the whole array (chains.py:276-300: one sample array per process). */
int gd_comm_rccl_path(char* buf, int32_t len, int32_t* preloaded_out);
int gd_upload_shard(gd_ctx* ctx, const double* X_cols, int64_t N, int64_t n, int64_t col_first, int64_t col_count,
                    int64_t col_stride, const double* weights);
int gd_comm_share_columns(gd_ctx* ctx, const int64_t* first_by_rank /* world + 1 entries */);
int gd_comm_unique_id(void* id128_out);
int gd_comm_init(gd_ctx* ctx, int32_t world, int32_t rank, const void* id128);
int gd_comm_info(gd_ctx* ctx, int32_t* world_out, int32_t* rank_out);
/* gd_comm_abandon: the one entry point that may be called from ANOTHER thread while a gd_comm_* call of this context has not
 * returned (a binding-side watchdog giving up on gd_comm_init): the stuck call, should it come back, installs no communicator
 * and returns GD_ERR_TIMEOUT.  (The reference has no multi-process path; this belongs to the build's RCCL seam) */
int gd_comm_abandon(gd_ctx* ctx);
int gd_comm_destroy(gd_ctx* ctx);
int gd_comm_allgather(gd_ctx* ctx, const double* send, int64_t count, double* recv);
int gd_comm_allreduce_sum(gd_ctx* ctx, double* inout, int64_t count);
int gd_comm_allgather_dev(gd_ctx* ctx, const void* d_send, int64_t count, void* d_recv);
int gd_comm_allreduce_sum_dev(gd_ctx* ctx, const void* d_send, int64_t count, void* d_recv);

#ifdef __cplusplus
}
#endif
#endif /* GDHIP_H */
