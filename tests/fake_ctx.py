"""
TEST DOUBLE of getdist_amd._lib.Context for the CPU (`-m "not gpu"`) tier.

It answers every Context call the host layer makes (getdist_amd/mcsamples.py, bench.one_step) with numpy / the
oracle, so the host logic -- ranges and limits, branch selection, batching, grouping by grid/frame class, the
asynchronous TNC pool, result assembly, the multi-rank partition -- can be exercised without a GPU.  It is test
infrastructure only: nothing in the product imports it, and it is NOT a fallback (MCSamples only uses it when a test
passes `_context_factory=`).
"""

import numpy as np
from scipy import fftpack

from oracle import kde_oracle as ko

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import planned_route  # noqa: E402

planned_route.install()  # this double has no gd_density2d_batch: its 2D batches take the Python-planned comparison route


class FakeBuf:
    def __init__(self, arr=None, nbytes=0):
        self.a = arr
        self.nbytes = nbytes
        self.ptr = id(self)

    def free(self):
        self.a = None

    def to_host(self, shape, dtype=np.float64, offset_bytes=0, pinned=False):
        return np.array(self.a, dtype=dtype).reshape(shape).copy()

    def to_host_async(self, shape, dtype=np.float64):
        return self.to_host(shape, dtype)

    def from_host(self, arr, offset_bytes=0):
        self.a = np.array(arr)


class FakeContext:
    calls = None  # class-level call log for tests that count launches

    def __init__(self, device=0):
        self.device = device
        self.N = self.n = 0
        self.log = []

    # ---- plumbing
    def close(self):
        pass

    def sync(self):
        pass

    def copy_sync(self):
        pass

    def copy_mark(self):  # the double completes every copy at once, so the lazy-delivery code runs on it unchanged
        return 0

    def copy_wait(self, token):
        pass

    def reserve_pinned_twin(self):
        pass

    def timer_start(self):
        pass

    def timer_stop_ms(self):
        return 0.0

    def alloc(self, nbytes):
        return FakeBuf(None, nbytes)

    def device_info(self):
        return dict(cu_count=1, lds_bytes=0, hbm_total=0, hbm_free=0, clock_khz=0, wave=64)

    def gather_items(self, dst, src, index, item_bytes, dst_offset=0):
        picked = np.array(src.a)[np.asarray(index, dtype=int)]
        if dst_offset == 0 and (dst.a is None or dst.nbytes <= len(index) * item_bytes):
            dst.a = picked
            return
        if dst.a is None or len(dst.a) < dst.nbytes // item_bytes:
            grown = np.zeros((dst.nbytes // item_bytes,) + picked.shape[1:])
            if dst.a is not None:
                grown[:len(dst.a)] = dst.a
            dst.a = grown
        dst.a[dst_offset:dst_offset + len(index)] = picked

    # ---- sample set
    def upload(self, samples, weights=None):
        self.s = np.asarray(samples, dtype=np.float64)
        if self.s.ndim == 1:
            self.s = self.s.reshape(-1, 1)
        self.N, self.n = self.s.shape
        self.s = np.hstack([self.s, np.zeros((self.N, self.EXTRA_COLS))])  # spare columns (gd_set_extra_column)
        self.w = None if weights is None else np.asarray(weights, dtype=np.float64)
        self.weighted = self.w is not None
        self._like_w, self._w_sel = None, 0

    def upload_shard(self, cols_f, N, n, col_first, weights=None):
        """gd_upload_shard: this rank's block of columns only; the others are NaN until comm_share_columns."""
        block = np.asarray(cols_f, dtype=np.float64).reshape(N, -1)
        self.N, self.n = int(N), int(n)
        self.s = np.full((self.N, self.n + self.EXTRA_COLS), np.nan)
        self.s[:, self.n:] = 0.0
        self.s[:, col_first:col_first + block.shape[1]] = block
        self.w = None if weights is None else np.asarray(weights, dtype=np.float64)
        self.weighted = self.w is not None
        self._like_w, self._w_sel = None, 0

    def comm_share_columns(self, first_by_rank):
        """gd_comm_share_columns: rank r broadcasts its block; ``comm_broadcast(array, root)`` is supplied by the test."""
        first = np.asarray(first_by_rank, dtype=np.int64)
        for r in range(len(first) - 1):
            a, b = int(first[r]), int(first[r + 1])
            if b > a:
                self.s[:, a:b] = self.comm_broadcast(np.ascontiguousarray(self.s[:, a:b]), r)

    def _w(self, lo=0, hi=None):
        hi = self.N if hi is None else hi
        if getattr(self, "_w_sel", 0):
            return self._like_w[lo:hi]
        return np.ones(hi - lo) if self.w is None else self.w[lo:hi]

    def density2d_masked(self, d_hist, pos, F, rx, ry, corr, winw, flags, bco, mbc, mask_bc, mask_mbc, zero_mask):
        """mcsamples.py:1884-1979 for one pair with explicit (already edge-masked) prior masks."""
        hist = np.asarray(d_hist.a)[pos]
        w_ = int(winw)
        Cinv = np.linalg.inv(np.array([[ry ** 2, rx * ry * corr], [rx * ry * corr, rx ** 2]]))
        i1, i2 = np.mgrid[-w_:w_ + 1, -w_:w_ + 1]
        Win = np.exp(-(i1**2 * Cinv[0, 0] + i2**2 * Cinv[1, 1] + 2 * Cinv[1, 0] * i1 * i2) / 2)
        Win /= np.sum(Win)
        big = F + 4 * w_ + 1
        px, py = bool(int(flags) & 16), bool(int(flags) & 32)  # histogram side circular on periodic axes (:1874-1881)
        mode = "periodic_both" if px and py else "periodic_x" if px else "periodic_y" if py else "same"
        bins2D = ko.conv2d(hist, Win, mode, largest_size=big)
        if bco >= 0 and not (px and py):
            a00 = ko.conv2d(mask_bc, Win, "valid", largest_size=big)
            ix = a00 * bins2D > np.max(bins2D) * 1e-8
            a00 = a00[ix]
            normed = bins2D[ix] / a00
            if bco == 0:
                bins2D[ix] = normed
            else:
                idx = np.arange(-w_, w_ + 1)
                y = np.repeat(idx[:, None], Win.shape[1], axis=1)
                winx, winy = Win * idx, Win * y
                a10 = ko.conv2d(mask_bc, winx, "valid", largest_size=big)[ix]
                a01 = ko.conv2d(mask_bc, winy, "valid", largest_size=big)[ix]
                a20 = ko.conv2d(mask_bc, winx * idx, "valid", largest_size=big)[ix]
                a02 = ko.conv2d(mask_bc, winy * y, "valid", largest_size=big)[ix]
                a11 = ko.conv2d(mask_bc, winy * idx, "valid", largest_size=big)[ix]
                xP = ko.conv2d(hist, winx, mode, largest_size=big)[ix]
                yP = ko.conv2d(hist, winy, mode, largest_size=big)[ix]
                denom = a20 * a01**2 + a10**2 * a02 - a00 * a02 * a20 + a11**2 * a00 - 2 * a01 * a10 * a11
                with np.errstate(divide="ignore", invalid="ignore"):
                    corrected = (bins2D[ix] * (a11**2 - a02 * a20) + xP * (a10 * a02 - a01 * a11) + yP * (a01 * a20 - a10 * a11)) / denom
                    bins2D[ix] = normed * np.exp(np.minimum(corrected / normed, 4) - 1)
        zero = np.asarray(zero_mask, dtype=bool)
        if mbc and not (px and py):
            a00 = ko.conv2d(mask_mbc, Win, "valid", largest_size=big)
            for _ in range(mbc):
                box = hist.copy()
                ix2 = bins2D > np.max(bins2D) * 1e-8
                box[ix2] /= bins2D[ix2]
                bins2D *= ko.conv2d(box, Win, mode, largest_size=big)
                bins2D[~zero] /= a00[~zero]
        bins2D[zero] = 0
        mx = np.max(bins2D)
        status = np.zeros(1, dtype=np.int32)
        if mx == 0:
            status[0] = -4
            return FakeBuf(np.zeros((1, F, F))), status
        return FakeBuf((bins2D / mx)[None]), status

    # ---- second lane
    def attach(self, owner):
        self.s, self.w, self.N, self.n, self.weighted = owner.s, owner.w, owner.N, owner.n, owner.weighted
        self._like_w, self._w_sel = None, 0

    def bind_thread(self):
        pass

    # ---- contour levels
    def contour_levels(self, d_P, B, F, contours):
        P = np.asarray(d_P.a).reshape(B, F, F)
        return np.array([ko.contour_levels(P[b], tuple(contours)) for b in range(B)]), np.zeros(B, dtype=np.int32)

    def isj1d(self, hist, neff):
        hist = np.asarray(hist, dtype=float)
        h = np.zeros(len(hist))
        status = np.zeros(len(hist), dtype=np.int32)
        for b in range(len(hist)):
            v = ko.isj_bandwidth_binned(hist[b], neff[b])
            if v is None:
                status[b] = -5
            else:
                h[b] = v
        return h, status

    def limits1d(self, P, x0, spacing, contours, factor=0):
        P = np.asarray(P, dtype=float)
        B, F = P.shape
        out = np.zeros((B, len(contours), 4))
        for b in range(B):
            x = x0[b] + spacing[b] * np.arange(F)
            out[b] = ko.density_limits_1d(x, P[b], contours, factor or None)
        return out, np.zeros(B, dtype=np.int32)

    # ---- thinned chains
    def weights_integral(self):
        w = self.w
        return w is None or bool(np.all(w == np.floor(w)) and np.all(w >= 0))

    def thin_rows(self, lo, hi, factor, unique_mode, capacity):
        from oracle import convergence_oracle as co

        w = np.ones(hi - lo) if self.w is None else self.w[lo:hi]
        assert bool(unique_mode) == (factor >= np.max(w))
        ix = co.thin_indices(factor, w) + lo
        assert len(ix) <= capacity
        return FakeBuf(ix.astype(np.int64)), len(ix)

    def binary_transitions(self, cols, rows, K, thresholds):
        thresholds = np.asarray(thresholds, dtype=float).reshape(len(cols), -1)
        out = np.zeros((len(cols), thresholds.shape[1], 12), dtype=np.int64)
        r = rows.a[:K]
        for ci, c in enumerate(cols):
            x = self.s[r, c]
            for t in range(thresholds.shape[1]):
                b = np.where(x >= thresholds[ci, t], 0, 1)
                out[ci, t, :8] = np.bincount(b[:-2] * 4 + b[1:-1] * 2 + b[2:], minlength=8)
                out[ci, t, 8:] = np.bincount(b[:-1] * 2 + b[1:], minlength=4)
        return out

    def thinned_lag_sums(self, cols, means, rows, K, maxoff):
        r = rows.a[:K]
        out = np.zeros((len(cols), maxoff))
        for ci, c in enumerate(cols):
            d = self.s[r, c] - means[ci]
            for off in range(1, maxoff + 1):
                out[ci, off - 1] = np.dot(d[off:], d[:-off])
        return out

    # ---- auxiliary vectors
    EXTRA_COLS = 4

    def set_extra_column(self, slot, x):
        assert 0 <= slot < self.EXTRA_COLS
        self.s[:, self.n + slot] = np.asarray(x, dtype=np.float64)
        return self.n + slot

    def aux_weights(self, w):
        assert not self._w_sel
        self._like_w = np.array(w, dtype=np.float64)

    def col_minmax(self, cols, lo=0, hi=None, cond_col=-1, cond_below=0.0):
        hi = self.N if hi is None else hi
        keep = self.s[lo:hi, cond_col] < cond_below if cond_col >= 0 else np.ones(hi - lo, dtype=bool)
        return np.array([[self.s[lo:hi, c][keep].min(), self.s[lo:hi, c][keep].max()] if keep.any()
                         else [np.inf, -np.inf] for c in cols])

    # ---- mean likelihoods
    def like_weights(self, loglikes, mode, mean_loglike):
        assert not getattr(self, "_w_sel", 0)
        if loglikes is None:
            self._like_w = None
            return None
        w = self._w()
        ll = np.asarray(loglikes, dtype=np.float64)
        self._like_w = w * ll if mode == 1 else w * np.exp(mean_loglike - ll)
        return float(np.sum(self._like_w))

    def select_weights(self, which):
        assert which in (0, 1) and (which == 0 or self._like_w is not None)
        self._w_sel = which

    def likes1d(self, hist, likehist, P, smooth, winw, flags, shade_mean_loglikes):
        hist, likehist, P = (np.asarray(a, dtype=float) for a in (hist, likehist, P))
        out = np.zeros_like(hist)
        status = np.zeros(len(hist), dtype=np.int32)
        for b in range(len(hist)):
            w_ = int(winw[b])
            x = np.arange(-w_, w_ + 1)
            Win = np.exp(-((x / smooth[b]) ** 2) / 2)
            Win /= np.sum(Win)
            mode = "periodic" if flags[b] & 4 else "same"
            raw = ko.conv1d(hist[b], Win, mode)
            ix = P[b] > 0
            fine = likehist[b].copy()
            fine[ix] /= P[b][ix]
            lk = ko.conv1d(fine, Win, mode)
            lk[ix] *= P[b][ix] / raw[ix]
            if shade_mean_loglikes:
                mn = np.min(lk)
                lk = np.where((lk - mn) < 30, np.exp(-(lk - mn)), 0)
                lk[raw == 0] = 0
            mx = np.max(lk)
            if mx == 0:
                status[b] = -4
            else:
                out[b] = lk / mx
        return out, status

    def likes2d(self, d_hist, d_likehist, B, F, rx, ry, corr, winw, flags, mbc):
        H = np.asarray(d_hist.a).reshape(B, F, F)
        LH = np.asarray(d_likehist.a).reshape(B, F, F)
        out = np.zeros((B, F, F))
        status = np.zeros(B, dtype=np.int32)
        for b in range(B):
            fl, w_ = int(flags[b]), int(winw[b])
            Cinv = np.linalg.inv(np.array([[ry[b] ** 2, rx[b] * ry[b] * corr[b]], [rx[b] * ry[b] * corr[b], rx[b] ** 2]]))
            i1, i2 = np.mgrid[-w_:w_ + 1, -w_:w_ + 1]
            Win = np.exp(-(i1**2 * Cinv[0, 0] + i2**2 * Cinv[1, 1] + 2 * Cinv[1, 0] * i1 * i2) / 2)
            Win /= np.sum(Win)
            px, py = bool(fl & 16), bool(fl & 32)
            mode = "periodic_both" if px and py else "periodic_x" if px else "periodic_y" if py else "same"
            L = ko.mean_likes_2d(H[b], LH[b], Win, mode, mbc, ko.conv2d_direct)  # direct summation, like the kernel
            if not np.max(L) > 0:
                status[b] = -4
            else:
                out[b] = L
        return FakeBuf(out), status

    # ---- moments
    def weight_stats(self, lo=0, hi=None, thresh=np.inf):
        w = self._w(lo, hi)
        return dict(norm=float(np.sum(w)), max_w=float(np.max(w)), sum_w2=float(np.dot(w, w)), n_above=float(np.sum(w > thresh)))

    def col_stats(self, lo=0, hi=None):
        hi = self.N if hi is None else hi
        s, w = self.s[lo:hi, :self.n], self._w(lo, hi)
        norm = np.sum(w)
        means = w.dot(s) / norm
        var = np.array([w.dot((s[:, i] - means[i]) ** 2) / norm for i in range(self.n)])
        return np.column_stack([s.min(axis=0), s.max(axis=0), means, var])

    def cov(self, cols=None, lo=0, hi=None, minmax=False):
        hi = self.N if hi is None else hi
        cols = list(range(self.n)) if cols is None else list(cols)
        s, w = self.s[lo:hi][:, cols], self._w(lo, hi)
        norm = np.sum(w)
        means = w.dot(s) / norm
        d = s - means
        out = (means, (d * w[:, None]).T @ d / norm, float(norm))
        return out + (np.column_stack([s.min(axis=0), s.max(axis=0)]),) if minmax else out

    def quantiles(self, cols, targets, lo=0, hi=None, minmax=None):
        hi = self.N if hi is None else hi
        targets = np.asarray(targets, dtype=float).reshape(len(cols), -1)
        out = np.zeros_like(targets)
        w = self._w(lo, hi)
        for r, c in enumerate(cols):
            x = self.s[lo:hi, c]
            idx = x.argsort()
            cum = np.cumsum(w[idx])
            out[r] = x[idx[np.minimum(np.searchsorted(cum, targets[r]), len(idx) - 1)]]
        return out

    def quantiles_probe(self, cols, targets, minmax, means):
        """gd_quantiles_mm_probe: the select plus the first 8 autocovariance lag sums of the same columns"""
        return self.quantiles(cols, targets, minmax=minmax), self.autocov_lags_batch(cols, means, 0, 8)

    # ---- lag sums
    # ---- stand-alone convolutions / likelihood statistics (numpy statements of the four entry points)
    def circ_convolve(self, a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        if a.ndim == 1:
            return np.fft.irfft(np.fft.rfft(a) * np.fft.rfft(b), n=a.size)
        return np.fft.irfftn(np.fft.rfftn(a) * np.fft.rfftn(b), a.shape, axes=(0, 1))

    def convolve1d_direct(self, x, y):
        return np.convolve(x, y, "full")

    def autoconvolve(self, s, n, normalize=True, x=None, col=-1, mean=0.0, use_weights=False):
        if x is None:
            x = (self.s[:, col] - mean) * (self._w() if use_weights else 1.0)
        x = np.asarray(x, dtype=np.float64)
        f = np.fft.rfft(x, s)
        r = np.fft.irfft((f * f.conj()).real, s)[:n]
        return r / np.arange(x.size, x.size - n, -1) if normalize else r

    def like_stats(self, col):
        L, w = self.s[:, col], self._w()
        mn = float(np.min(L))
        return dict(min=mn, max=float(np.max(L)), norm=float(np.sum(w)), sum_wl=float(np.dot(w, L)),
                    sum_wl2=float(np.dot(w, L * L)), sum_w_exp_plus=float(np.dot(w, np.exp(L - mn))),
                    sum_w_exp_minus=float(np.dot(w, np.exp(-(L - mn)))), argmin=int(np.argmin(L)))

    def autocov_lags_range_batch(self, cols, means, lo, hi, k0, nlags):
        out = np.zeros((len(cols), nlags))
        w = self._w(lo, hi)
        for r, (c, m) in enumerate(zip(cols, means)):
            d = (self.s[lo:hi, c] - m) * w
            for l in range(nlags):
                k = k0 + l
                out[r, l] = np.dot(d[:len(d) - k], d[k:]) if k < len(d) else 0.0
        return out

    def autocov_lags_batch(self, cols, means, k0, nlags):
        return self.autocov_lags_range_batch(cols, means, 0, self.N, k0, nlags)

    def autocov_lags(self, col, mean, k0, nlags):
        return self.autocov_lags_batch([col], [mean], k0, nlags)[0]

    def kde_lag_sums_batch(self, cols, inv4s2, lags):
        w = self._w()
        out = np.zeros((len(cols), len(lags)))
        for r, (c, cc) in enumerate(zip(cols, inv4s2)):
            x = self.s[:, c]
            for q, k in enumerate(lags):
                out[r, q] = np.dot(np.exp(-((x[:-k] - x[k:]) ** 2) * cc) * w[:-k], w[k:])
        return out

    def kde_lag_sums(self, col, inv4s2, lags):
        return self.kde_lag_sums_batch([col], [inv4s2], lags)[0]

    def kde_lag_sums_2d(self, coli, colj, kinv3, lags):
        w = self._w()
        x, y = self.s[:, coli], self.s[:, colj]
        out = np.zeros(len(lags))
        for q, k in enumerate(lags):
            dx, dy = x[:-k] - x[k:], y[:-k] - y[k:]
            out[q] = np.dot(np.exp(-(kinv3[0] * dx * dx + kinv3[1] * dx * dy + kinv3[2] * dy * dy) / 4) * w[:-k], w[k:])
        return out

    # ---- binning
    def hist1d(self, cols, binmin, width, F):
        w = self._w()
        return np.array([np.bincount(((self.s[:, c] - b) / wd + 0.5).astype(int), weights=w, minlength=F)[:F]
                         for c, b, wd in zip(cols, binmin, width)])

    def prebin(self, col, binmin, width, F, buf=None):
        self.log.append(("prebin", col, F))
        return FakeBuf(((self.s[:, col] - binmin) / width + 0.5).astype(np.int64))

    def prebin_batch(self, cols, binmin, width, F, bufs):
        for c, b, wd, buf in zip(cols, binmin, width, bufs):
            buf.a = ((self.s[:, c] - b) / wd + 0.5).astype(np.int64)

    def hist2d_prebinned(self, idx_x, idx_y, F, out=None):
        self.log.append(("hist2d_prebinned", len(idx_x), F))
        w = self._w()
        H = np.array([np.bincount(bx.a + by.a * F, weights=w, minlength=F * F).reshape(F, F) for bx, by in zip(idx_x, idx_y)])
        return FakeBuf(H)

    def minmax_affine(self, coli, colj, a, b):
        return np.array([[np.min(aa * self.s[:, i] + bb * self.s[:, j]), np.max(aa * self.s[:, i] + bb * self.s[:, j])]
                         for i, j, aa, bb in zip(coli, colj, a, b)])

    def hist2d_sheared(self, coli, colj, r0, r1, xmin, dx, ymin, dy, F, out=None):
        w = self._w()
        H = []
        for i, j, a, b, x0, ddx, y0, ddy in zip(coli, colj, r0, r1, xmin, dx, ymin, dy):
            b1 = ((self.s[:, i] - x0) / ddx).astype(int)
            b2 = (((a * self.s[:, i] + b * self.s[:, j]) - y0) / ddy).astype(int)
            H.append(np.bincount(b1 + b2 * F, weights=w, minlength=F * F).reshape(F, F))
        return FakeBuf(np.array(H))

    # ---- 1D density
    def dct1d(self, hist):
        hist = np.asarray(hist, dtype=float)
        return np.array([fftpack.dct(h / np.sum(h)) for h in hist])

    def density1d(self, hist, smooth, winw, flags, bco, mbc):
        hist = np.asarray(hist, dtype=float)
        B, F = hist.shape
        P_out = np.zeros_like(hist)
        status = np.zeros(B, dtype=np.int32)
        for b in range(B):
            bins = hist[b]
            bot, top, periodic = bool(flags[b] & 1), bool(flags[b] & 2), bool(flags[b] & 4)
            kernel = ko.Kernel1D(int(winw[b]), smooth[b])
            w_ = int(winw[b])
            mode = "periodic" if periodic else "same"
            P = ko.conv1d(bins, kernel.Win, mode)
            if (bot or top) and not periodic and bco >= 0:
                mask = np.ones(F + 2 * w_)
                if bot:
                    mask[w_] = 0.5
                    mask[:w_] = 0
                if top:
                    mask[-(w_ + 1)] = 0.5
                    mask[-w_:] = 0
                a0 = ko.conv1d(mask, kernel.Win, "valid")
                ix = np.nonzero(a0 * P)
                a0 = a0[ix]
                normed = P[ix] / a0
                if bco == 0:
                    P[ix] = normed
                else:
                    xWin = kernel.Win * kernel.x
                    a1 = ko.conv1d(mask, xWin, "valid")[ix]
                    a2 = ko.conv1d(mask, xWin * kernel.x, "valid")[ix]
                    xP = ko.conv1d(bins, xWin, "same")[ix]
                    if bco == 1:
                        corrected = (P[ix] * a2 - xP * a1) / (a0 * a2 - a1**2)
                    else:
                        a3 = ko.conv1d(mask, xWin * kernel.x**2, "valid")[ix]
                        a4 = ko.conv1d(mask, xWin * kernel.x**3, "valid")[ix]
                        x2P = ko.conv1d(bins, xWin * kernel.x, "same")[ix]
                        denom = a4 * a2 * a0 - a4 * a1**2 - a2**3 - a3**2 * a0 + 2 * a1 * a2 * a3
                        corrected = (P[ix] * (a4 * a2 - a3**2) + xP * (a2 * a3 - a4 * a1) + x2P * (a3 * a1 - a2**2)) / denom
                    P[ix] = normed * np.exp(np.minimum(corrected / normed, 4) - 1)
            elif not periodic and bco == 2:
                xWin2 = kernel.Win * kernel.x**2
                x2P = ko.conv1d(bins, xWin2, "same")
                a2 = np.sum(xWin2)
                a4 = np.dot(xWin2, kernel.x**2)
                corrected = (P * a4 - a2 * x2P) / (a4 - a2**2)
                ix = P > 0
                P[ix] *= np.exp(np.minimum(corrected[ix] / P[ix], 2) - 1)
            if mbc:
                if not periodic:
                    m2 = np.ones(F)
                    if bot:
                        m2[0] *= 0.5
                    if top:
                        m2[-1] *= 0.5
                    a0 = ko.conv1d(m2, kernel.Win, "same")
                for _ in range(mbc):
                    p1 = P.copy()
                    p1[p1 == 0] = 1
                    P = P * ko.conv1d(bins / p1, kernel.Win, mode)
                    if not periodic:
                        P /= a0
            mx = np.max(P)
            if mx == 0:
                status[b] = -4
            else:
                P_out[b] = P / mx
        return P_out, status

    # ---- 2D
    def kopt2d(self, d_hist, B, F, neff, do_corr, fallback_t, corr):
        self.log.append(("kopt2d", B, F))
        H = np.asarray(d_hist.a).reshape(B, F, F)
        out = np.full((B, 12), np.nan)
        for b in range(B):
            tr = {}
            out[b, 11] = -5
            try:
                opt = ko.Optimizer2D(H[b], neff[b], corr[b], do_correlation=bool(do_corr[b]),
                                     fallback_t=(fallback_t[b] if fallback_t[b] > 0 else None), trace=tr)
                out[b, 7] = 0
                try:
                    out[b, 8:11] = opt.get_h()
                    out[b, 11] = 0
                except Exception:
                    out[b, 11] = -1  # "bias not positive definite"
                out[b, 0] = tr["t_star"]
                out[b, 1:4] = tr["p_02"], tr["p_20"], tr["p_11"]
                if do_corr[b]:
                    out[b, 4:7] = tr["p_00"], tr["p_13"], tr["p_31"]
            except ValueError:
                out[b, 7] = -5
        return out

    def get_h(self, psi, neff, corr, do_corr):
        psi = np.asarray(psi, dtype=float).reshape(-1, 6)
        out = np.zeros((len(psi), 4))
        for b in range(len(psi)):
            try:
                out[b, :3] = ko.get_h_from_psi(tuple(psi[b]), neff[b], corr[b], bool(do_corr[b]))
            except Exception:
                out[b, 3] = -1
        return out

    def pinned_array(self, shape, dtype=np.float64):
        return np.zeros(shape, dtype=dtype)

    def density2d_enqueue(self, d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc, status):
        out, st = self.density2d(d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc)
        status[:] = st
        return out

    def density2d(self, d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc, out=None):
        self.log.append(("density2d", B, F))
        H = np.asarray(d_hist.a).reshape(B, F, F)
        P_out = np.zeros((B, F, F))
        status = np.zeros(B, dtype=np.int32)

        class _P:  # stand-in for ParamState in the oracle's mask helpers
            def __init__(self, bot, top, periodic):
                self.has_limits_bot, self.has_limits_top, self.periodic = bot, top, periodic

        for b in range(B):
            fl, w_ = int(flags[b]), int(winw[b])
            parx = _P(bool(fl & 1), bool(fl & 2), bool(fl & 16))
            pary = _P(bool(fl & 4), bool(fl & 8), bool(fl & 32))
            has_prior = bool(fl & 64) or bool(fl & 15)
            Cinv = np.linalg.inv(np.array([[ry[b] ** 2, rx[b] * ry[b] * corr[b]], [rx[b] * ry[b] * corr[b], rx[b] ** 2]]))
            i1, i2 = np.mgrid[-w_:w_ + 1, -w_:w_ + 1]
            Win = np.exp(-(i1**2 * Cinv[0, 0] + i2**2 * Cinv[1, 1] + 2 * Cinv[1, 0] * i1 * i2) / 2)
            Win /= np.sum(Win)
            mode = ("periodic_both" if parx.periodic and pary.periodic else "periodic_x" if parx.periodic
                    else "periodic_y" if pary.periodic else "same")
            hist = H[b]
            big = F + 4 * w_ + 1
            bins2D = ko.conv2d(hist, Win, mode, largest_size=big)
            both = parx.periodic and pary.periodic
            mask = np.ones((F + 2 * w_, F + 2 * w_))
            if has_prior and bco >= 0 and not both:
                ko._set_edge_mask_2d(parx, pary, mask, w_)
                a00 = ko.conv2d(mask, Win, "valid", largest_size=big)
                ix = a00 * bins2D > np.max(bins2D) * 1e-8
                a00 = a00[ix]
                normed = bins2D[ix] / a00
                if bco == 0:
                    bins2D[ix] = normed
                else:
                    idx = np.arange(-w_, w_ + 1)
                    y = np.repeat(idx[:, None], Win.shape[1], axis=1)
                    winx, winy = Win * idx, Win * y
                    a10 = ko.conv2d(mask, winx, "valid", largest_size=big)[ix]
                    a01 = ko.conv2d(mask, winy, "valid", largest_size=big)[ix]
                    a20 = ko.conv2d(mask, winx * idx, "valid", largest_size=big)[ix]
                    a02 = ko.conv2d(mask, winy * y, "valid", largest_size=big)[ix]
                    a11 = ko.conv2d(mask, winy * idx, "valid", largest_size=big)[ix]
                    xP = ko.conv2d(hist, winx, mode, largest_size=big)[ix]
                    yP = ko.conv2d(hist, winy, mode, largest_size=big)[ix]
                    denom = a20 * a01**2 + a10**2 * a02 - a00 * a02 * a20 + a11**2 * a00 - 2 * a01 * a10 * a11
                    corrected = (bins2D[ix] * (a11**2 - a02 * a20) + xP * (a10 * a02 - a01 * a11) + yP * (a01 * a20 - a10 * a11)) / denom
                    bins2D[ix] = normed * np.exp(np.minimum(corrected / normed, 4) - 1)
            if mbc and not both:
                ko._set_all_edge_mask_2d(mask, w_, parx.periodic, pary.periodic)
                a00 = ko.conv2d(mask, Win, "valid", largest_size=big)
                for _ in range(mbc):
                    box = hist.copy()
                    ix2 = bins2D > np.max(bins2D) * 1e-8
                    box[ix2] /= bins2D[ix2]
                    bins2D *= ko.conv2d(box, Win, mode, largest_size=big)
                    bins2D /= a00
            mx = np.max(bins2D)
            if mx == 0:
                status[b] = -4
            else:
                P_out[b] = bins2D / mx
        return FakeBuf(P_out), status
