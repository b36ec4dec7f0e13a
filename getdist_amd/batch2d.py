"""
Python side of gd_density2d_batch (include/gdhip.h): packs what MCSamples knows -- settings, the per-parameter state
_initParam left, the correlation / covariance matrices, the pair list -- into the structures of the C ABI, makes ONE
call, and wraps the page-locked grids it filled into Density2D objects.  Every decision of get2DDensityGridData /
getAutoBandwidth2D (mcsamples.py:1285-1419, 1748-2010 of the reference) is taken inside the library
(getdist_amd/csrc/batch2d.hpp); nothing here computes.
"""

import ctypes as C
import functools
import logging
import threading
from collections.abc import Sequence

import numpy as np

from .densities import DensitiesError, Density2D

META = 32
NEED_NEFF = -20


class ParamState(C.Structure):
    """gd_param2d"""

    _fields_ = [("range_min", C.c_double), ("range_max", C.c_double), ("param_min", C.c_double), ("param_max", C.c_double),
                ("sigma_range", C.c_double), ("err", C.c_double), ("mean", C.c_double), ("var", C.c_double),
                ("neff", C.c_double), ("has_limits_bot", C.c_int32), ("has_limits_top", C.c_int32),
                ("periodic", C.c_int32), ("owned", C.c_int32)]


class BatchSettings(C.Structure):
    """gd_batch2d_settings"""

    _fields_ = [("fine_bins_2D", C.c_int32), ("boundary_correction_order", C.c_int32),
                ("mult_bias_correction_order", C.c_int32), ("num_bins_2D", C.c_int32),
                ("smooth_scale_2D", C.c_double), ("max_corr_2D", C.c_double), ("norm", C.c_double), ("sum_w2", C.c_double),
                ("uncorrelated_sampler", C.c_int32), ("raise_on_bandwidth_errors", C.c_int32),
                ("want_levels", C.c_int32), ("ncontours", C.c_int32), ("contours", C.POINTER(C.c_double)),
                ("two_streams_min", C.c_int32), ("two_streams_split", C.c_int32), ("kopt_split_min", C.c_int32),
                ("kopt_first_fraction", C.c_double), ("first_batch", C.c_int32), ("max_batch", C.c_int32),
                ("max_batch_bytes", C.c_double), ("comm_exchange", C.c_int32), ("bandwidths_only", C.c_int32),
                ("pair_neff", C.POINTER(C.c_double)), ("bandwidths", C.POINTER(C.c_double)),
                ("results_in_flight", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


class PendingBatch:
    """The tail of a native batched call: its result copies may still be in flight.  wait() blocks until THIS call's
    copies have landed (marks on the copy streams of the one or two contexts), idempotent and serialised by a lock;
    wait_grid(k) -- a grid's first read -- completes the call and raises DensitiesError on every read of a grid that came
    back empty.  The device blocks belong to the library, which releases them when the next call completes this one."""

    def __init__(self, ctxs, tokens, status, status_index):
        self.ctxs, self.tokens = ctxs, tokens
        self.status, self.status_index = status, status_index
        self.done = False
        self.failed = frozenset()
        self.lock = threading.Lock()

    def wait(self):
        with self.lock:
            if self.done:
                return
            for c, tok in zip(self.ctxs, self.tokens):
                if tok >= 0 and getattr(c, "h", True) is not None:  # a closed context has synchronised on the way out
                    c.copy_wait(tok)
            bad = np.nonzero(np.asarray(self.status)[self.status_index] != 0)[0]
            self.failed = frozenset(bad.tolist())
            self.done = True

    def wait_grid(self, k):
        self.wait()
        if k in self.failed:
            raise DensitiesError("no samples in bin")


class DensityBatch(Sequence):
    """
    The results of one native batched call: a sequence of Density2D that are made when they are first asked for
    (``collections.abc.Sequence``: indexing, slicing -- slices are lists --, iteration, ``len``, ``in``; ``+`` and ``*`` give
    plain lists; it is not a ``list`` instance: ``list(out)`` makes one).  The
    grids are views of the call's page-locked block (filled by the copy streams while the caller goes on), the axes, the
    bandwidth records and the contour levels come from the call's per-pair table; building 1225 result objects costs a few
    milliseconds of interpreter time, which a caller that loops over batched calls would otherwise spend between the last
    enqueue of one call and the first launch of the next.
    """

    def __init__(self, mc, pairs32, F_v, meta, grids, completion, levels, level_status, contours, auto):
        self._names = mc.paramNames.names
        self._view = [(getattr(p, "range_min", None), getattr(p, "range_max", None)) for p in self._names]  # as of this call
        self._all_contours = mc.contours
        self._n = mc.n
        self._pairs, self._F, self._meta, self._grids = pairs32, F_v, meta, grids
        self._completion, self._levels, self._level_status, self._contours, self._auto = completion, levels, level_status, contours, auto
        self._items = [None] * len(pairs32)
        self._ax = None

    def __len__(self):
        return len(self._items)

    def _axes(self):
        """The grid axes of every (parameter, F) in use: np.linspace(lo, hi, F) written out on one 2D array per F."""
        P = len(self._items)
        meta, F_v = self._meta, self._F
        jx, jy = self._pairs[:, 0], self._pairs[:, 1]
        lo_all, hi_all = np.zeros(self._n), np.zeros(self._n)  # binmin / binmax per column (the same for every grid size)
        lo_all[jx], hi_all[jx] = meta[:P, 23], meta[:P, 24]
        lo_all[jy], hi_all[jy] = meta[:P, 25], meta[:P, 26]
        ax = {}
        for F_ in np.unique(F_v).tolist():
            sel = F_v == F_
            js_ = np.unique(np.concatenate([jx[sel], jy[sel]]))
            lo_, hi_ = lo_all[js_], hi_all[js_]
            A = np.arange(F_, dtype=np.float64)[None, :] * ((hi_ - lo_) / (F_ - 1))[:, None] + lo_[:, None]
            A[:, -1] = hi_
            for row, j in enumerate(js_.tolist()):
                ax[(j, F_)] = (A[row], A[row, 1] - A[row, 0], self._view[j])
        self._ax = ax
        return ax

    def _make(self, k):
        ax_cache = self._ax or self._axes()
        meta = self._meta
        F = int(self._F[k])
        j, j2 = int(self._pairs[k, 0]), int(self._pairs[k, 1])
        ax, sx, vrx = ax_cache[(j, F)]
        ay, sy, vry = ax_cache[(j2, F)]
        off = int(meta[k, 1])
        Pk = self._grids[off:off + F * F].reshape(F, F)
        cont = None
        state = None if self._level_status is None else int(self._level_status[k])
        if state == 0:
            cont = self._levels[k].copy()
        auto = self._auto
        dens = Density2D._from_fields(dict(
            x=ax, y=ay, axes=[ay, ax], spacing=sx * sy, view_ranges=[vrx, vry], mask=None, likes=None, contours=cont, spl=None,
            _P=Pk, _wait=functools.partial(self._completion.wait_grid, k) if self._completion is not None else None,
            bandwidth=tuple(meta[k, 2:5].tolist()) if auto else None,
            bandwidth_branch="ABC"[int(meta[k, 5])] if auto and meta[k, 5] >= 0 else None,  # (-1: injected bandwidths)
            kopt=None if np.isnan(meta[k, 13]) else meta[k, 6:18].copy()))
        if cont is None and state is not None:  # more exactly equal grid values at the level than the kernel's tie list holds
            dens.contours = dens.getContourLevels(self._all_contours[:len(self._contours)])
        return dens

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[q] for q in range(*k.indices(len(self._items)))]
        k = int(k)
        if k < 0:
            k += len(self._items)
        d = self._items[k]
        if d is None:
            d = self._items[k] = self._make(k)
        return d

    def __iter__(self):
        for k in range(len(self._items)):
            yield self[k]

    def __reduce__(self):  # pickles / copies as the plain list of its densities
        return (list, (list(self),))

    def __eq__(self, other):
        return list(self) == list(other)

    __hash__ = None  # (a mutable sequence of results, like the list the Python-planned route returns: unhashable)

    # list arithmetic, so that callers written against the list the reference's triangle loop builds keep working
    # (`out + other_list`, `other_list + out`, `out * 2`): the result is a plain list
    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __mul__(self, k):
        return list(self) * k

    __rmul__ = __mul__


def settings_of(mc, base_F, bco, mbc, smooth_scale_2D, want_levels, contours):
    s = BatchSettings()
    s.fine_bins_2D, s.boundary_correction_order, s.mult_bias_correction_order = int(base_F), int(bco), int(mbc)
    s.num_bins_2D = int(mc.num_bins_2D)
    s.smooth_scale_2D, s.max_corr_2D = float(smooth_scale_2D), float(mc.max_corr_2D)
    s.norm, s.sum_w2 = float(mc.norm), float(mc._sum_w2)
    s.uncorrelated_sampler = int(mc.sampler in ("nested", "uncorrelated"))
    s.raise_on_bandwidth_errors = int(bool(mc.raise_on_bandwidth_errors))
    s.want_levels = int(bool(want_levels))
    if want_levels:
        s.ncontours = len(contours)
        s.contours = contours.ctypes.data_as(C.POINTER(C.c_double))
    s.two_streams_min, s.two_streams_split = mc.CONV_TWO_STREAMS_PAIRS
    s.kopt_split_min, s.kopt_first_fraction = int(mc.KOPT_SPLIT_MIN), float(mc.KOPT_FIRST_FRACTION)
    return s


def pack_params(mc, used, owned=None):
    """gd_param2d records (index = column) of the parameters in ``used``; ``owned``: the columns whose N_eff this rank
    computes in a multi-rank job (None: all)."""
    names = mc.paramNames.names
    arr = (ParamState * mc.n)()
    means, vars_ = np.asarray(mc.means), np.asarray(mc.vars)
    for j in range(mc.n):
        arr[j].neff = np.nan if names[j].N_eff_kde is None else names[j].N_eff_kde
    for j in used:
        p, r = names[j], arr[j]
        r.range_min, r.range_max, r.param_min, r.param_max = p.range_min, p.range_max, p.param_min, p.param_max
        r.sigma_range = np.nan if p.sigma_range is None else p.sigma_range
        r.err, r.mean, r.var = p.err, means[j], vars_[j]
        r.neff = np.nan if p.N_eff_kde is None else p.N_eff_kde
        r.has_limits_bot, r.has_limits_top, r.periodic = bool(p.has_limits_bot), bool(p.has_limits_top), bool(p.periodic)
        r.owned = (1 if owned is None or j in owned else 0) | (2 if mc._no_bandwidth_warning(p) else 0)
    return arr


def run(mc, pa, base_F, bco, mbc, smooth_scale_2D, num_plot_contours, get_density, bandwidths=None, pair_neff=None,
        bandwidths_only=False):
    """The native route of MCSamples.get2DDensities: ``pa`` is the (P, 2) int array of column pairs.
    ``bandwidths`` (P x 3: hx, hy, corr in parameter units) replaces getAutoBandwidth2D, ``pair_neff`` (P) the effective
    sample numbers it would use (use_effective_samples_2D); ``bandwidths_only`` returns the call's per-pair table
    (``meta``, include/gdhip.h) and the grid sizes once the bandwidths are known, without convolving anything."""
    from . import mcsamples as M
    from ._lib import GdhipError

    ctx = mc.ctx
    names = mc.paramNames.names
    P = len(pa)
    pairs32 = np.ascontiguousarray(pa, dtype=np.int32).reshape(-1, 2)
    flat = pairs32.ravel()
    used = flat[np.sort(np.unique(flat, return_index=True)[1])].tolist() if P else []
    share = getattr(mc, "_neff_share", None)
    if P == 0:
        if share is not None:
            share.complete(mc)  # a rank without pairs still enters the step's collective
        return []
    mc._init_params(used)
    contours = None
    if not get_density:
        ncontours = len(mc.contours)
        if num_plot_contours:
            ncontours = min(num_plot_contours, ncontours)
        contours = np.ascontiguousarray(mc.contours[:ncontours], dtype=np.float64)
    settings = settings_of(mc, base_F, bco, mbc, smooth_scale_2D, not get_density, contours)
    keep = []  # (arrays the settings point to)
    if smooth_scale_2D < 0 and bandwidths is not None:
        keep.append(np.ascontiguousarray(np.array(list(bandwidths), dtype=np.float64).reshape(P, 3)))
        settings.bandwidths = keep[-1].ctypes.data_as(C.POINTER(C.c_double))
    if smooth_scale_2D < 0 and pair_neff is not None:
        keep.append(np.ascontiguousarray(pair_neff, dtype=np.float64).reshape(P))
        settings.pair_neff = keep[-1].ctypes.data_as(C.POINTER(C.c_double))
    settings.bandwidths_only = int(bool(bandwidths_only))
    corr = np.ascontiguousarray(mc.getCorrelationMatrix(), dtype=np.float64)
    cov = np.ascontiguousarray(mc.getCov(), dtype=np.float64)
    # the autocovariance probe prepareParams may have started beside its quantile select
    lag_probe = None
    pre, mc._lag_prefetch = getattr(mc, "_lag_prefetch", None), None
    if pre is not None:
        cols, pnl, fut = pre
        try:
            lags = fut.result()
        except Exception:
            lags = None
        if lags is not None and pnl == 8:
            lag_probe = np.full((mc.n, 8), np.nan)
            lag_probe[cols] = lags
    twin = None
    if P >= min(64, mc.CONV_TWO_STREAMS_PAIRS[0]) and mc._context_factory is not None:
        nlanes = mc._nlanes
        twin = mc._second_lane().ctx
        mc._nlanes = nlanes
    F_v = ctx.batch2d_grid_sizes(settings, mc.n, corr, pairs32)
    total = int(np.sum(F_v.astype(np.int64) ** 2))
    # (size classes of an eighth of a power of two: calls of similar size recycle one another's page-locked blocks -- a fresh
    # gigabyte of page-locked memory costs ~0.2 s)
    gran = 1 << max(int(total).bit_length() - 4, 13)
    grids = ctx.pinned_array(((max(0 if bandwidths_only else total, 1) + gran - 1) // gran * gran,), np.float64)
    status = ctx.pinned_array((max(P, 1),), np.int32)
    meta = np.empty((max(P, 1), META))
    levels = level_status = None
    if not get_density:
        levels, level_status = np.zeros((max(P, 1), len(contours))), np.zeros(max(P, 1), dtype=np.int32)
    cb = None
    library_exchange = share is not None and not share.exchanged and getattr(share, "library_comm", False)
    settings.comm_exchange = int(library_exchange)
    if share is not None and not share.exchanged and not library_exchange:
        def exchange(user, neff_ptr, n):
            try:
                v = np.ctypeslib.as_array(neff_ptr, shape=(n,))
                for j in range(n):
                    if names[j].N_eff_kde is None and not np.isnan(v[j]):
                        names[j].N_eff_kde = float(v[j])
                share.exchange(mc)
                share.exchanged = True
                for j in range(n):
                    if names[j].N_eff_kde is not None:
                        v[j] = names[j].N_eff_kde
                return 0
            except Exception:  # (never let an exception unwind through the C frames)
                logging.exception("N_eff exchange failed")
                return 1

        cb = EXCHANGE_FN(exchange)
    previous = mc._pending_results
    # (a stream of calls: this call's first grids queue behind the previous call's copies -- the library schedules for
    # throughput then, for the delivery of this call's grids otherwise)
    settings.results_in_flight = int(previous is not None and not previous.done)
    # The library's own exchange (comm_exchange) may have been ENTERED by a call that failed afterwards -- 'Matrix is not
    # positive definite', a BandwidthError, a device error, NEED_NEFF from the columns nobody owned: the library counts the
    # collectives it issued (gd_batch2d_exchanges), and the flag follows the count in a `finally`, before the NEED_NEFF
    # retry and before any exception leaves -- so that neither the retry, nor NeffShare.complete() in the caller's error
    # path, enters a second all-reduce that no other rank matches (advisor finding, round 4).
    count0 = ctx.batch2d_exchanges() if library_exchange else 0

    def note_exchange():
        if library_exchange and not share.exchanged and ctx.batch2d_exchanges() > count0:
            share.exchanged = True

    for attempt in (0, 1):
        params = pack_params(mc, used, None if share is None else share.params)
        try:
            try:
                tokens = ctx.density2d_batch(twin, settings, params, mc.n, corr, cov, lag_probe, pairs32, cb, grids, status, meta,
                                             levels, level_status)
            finally:
                note_exchange()
            break
        except GdhipError as e:
            if e.code == NEED_NEFF and attempt == 0:
                # a chain whose correlation outlasts the 8-lag probe: getCorrelationLength's long route stays on the
                # Python side (growing chunks of lag sums, then the length-2N transform); the values are cached on the
                # parameters, so the second call finds them
                if share is not None and share.exchanged:
                    # the other ranks' values arrived with the collective this rank has already taken part in: keep them,
                    # so that _neff_batch's completion step does not look for an exchange
                    for j in range(mc.n):
                        if names[j].N_eff_kde is None and not np.isnan(params[j].neff):
                            names[j].N_eff_kde = float(params[j].neff)
                mc._neff_batch(used)
                cb = None if share is None or share.exchanged else cb
                settings.comm_exchange = int(library_exchange and not share.exchanged)
                continue
            msg = str(e)
            if "bias not positive definite" in msg:
                raise Exception("bias not positive definite")
            if e.code == -5:
                raise M.BandwidthError(msg.split(": ", 1)[-1])
            if e.code == -1:
                raise M.SettingError(msg.split(": ", 1)[-1])
            raise
    for j in (range(mc.n) if share is not None else used):
        if names[j].N_eff_kde is None and not np.isnan(params[j].neff):
            names[j].N_eff_kde = float(params[j].neff)
    warn = meta[:P, 22].astype(np.int64)
    for k in np.nonzero(warn & 1)[0].tolist():
        logging.warning("Parameters are 100%% correlated: %s, %s", names[pairs32[k, 0]].name, names[pairs32[k, 1]].name)
    for k in np.nonzero(warn & 4)[0].tolist():
        logging.warning("2D kernel density bandwidth optimizer failed for %s, %s. Using fallback width: %s",
                        names[pairs32[k, 0]].name, names[pairs32[k, 1]].name, "2D fixed point: no root in [0, 0.1]")
    for k in np.nonzero(warn & 2)[0].tolist():
        logging.warning("fine_bins_2D not large enough for optimal density: %s, %s", names[pairs32[k, 0]].name,
                        names[pairs32[k, 1]].name)
    if bandwidths_only:
        return meta[:P], F_v
    lazy = get_density
    ctxs = [ctx] + ([twin] if twin is not None else [])
    completion = PendingBatch(ctxs, list(tokens)[:len(ctxs)], status, meta[:P, 30].astype(np.int64))
    if not lazy:
        completion.wait()
        if level_status is not None and np.any(level_status[:P] == -4):
            raise DensitiesError("Contour level outside plotted ranges")
        if completion.failed:
            raise DensitiesError("no samples in bin")
    out = DensityBatch(mc, pairs32, F_v, meta, grids, completion if lazy else None, levels, level_status, contours,
                       smooth_scale_2D < 0)
    if not lazy:
        return out
    mc._pending_results = completion
    if previous is not None:
        previous.wait()  # (its copies are ahead of this call's on the copy streams)
    return out
