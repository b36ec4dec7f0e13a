"""
The Python-planned 2D pipeline: rounds 2-5's orchestration of the 2D kernels (binning, N_eff, branch plan, optimiser
launches, convolution batches, result copies) written out in Python over the context's entry points.  The product's
MCSamples.get2DDensities makes ONE native call (gd_density2d_batch, getdist_amd/csrc/batch2d.hpp) for every branch; this
sequence is kept here, under tests/, as the comparison the native route is held bit-equal to (tests/test_native_batch.py on
the numpy context double, tests/test_gpu_native_batch.py on the device with GETDIST_AMD_NATIVE_BATCH=0) and as the route
of contexts that have no native entry (tests/fake_ctx.py: the CPU suite's host-logic tests).

install() registers it as MCSamples._planned_route; tests/fake_ctx.py and tests/conftest.py call it.
Follows getdist/mcsamples.py:1748-2010 (get2DDensityGridData) and :1285-1419 (getAutoBandwidth2D) of the reference.
"""
from getdist_amd import mcsamples as _M

globals().update({k: v for k, v in vars(_M).items() if not k.startswith("__")})  # the module's helpers, by their names


class _PendingResults:
    """
    The tail of a batched 2D call whose result copies are still in flight: the device grids, the inputs that must not
    be recycled before the kernels reading them have run, and the page-locked status words.

    wait() blocks until THIS call's copies have landed (a mark on the copy stream, so a later call's copies are not
    waited for), releases the device blocks and records which grids came back empty; it is idempotent, serialised by a
    lock (a second reader thread returns only after the copies have landed), and marks itself done only after the wait
    succeeded.  wait_grid(k) -- what a grid's first read calls -- completes the call and raises DensitiesError every
    time the grid of pair ``k`` is read if THAT grid had no samples; its siblings are unaffected.  The next batched
    call on the same object completes this one before it takes its place, so an unread triangle never pins its device
    blocks beyond one further call.
    """

    def __init__(self, ctx, inflight, release, more_ctxs=()):
        self.ctx, self.inflight, self.release = ctx, inflight, release
        self.ctxs = [ctx] + list(more_ctxs)  # a small call spreads its batches over the streams of two contexts
        self.mark()
        self.done = False
        self.failed = frozenset()
        self.lock = threading.Lock()

    def mark(self):
        """'Every result copy issued so far' on each context's copy stream."""
        self.tokens = [c.copy_mark() for c in self.ctxs]
        self.token = self.tokens[0]

    def wait(self):
        with self.lock:
            if self.done:
                return
            for c, tok in zip(self.ctxs, self.tokens):
                if getattr(c, "h", True) is not None:  # a closed context has synchronised its streams on the way out
                    c.copy_wait(tok)
            failed = set()
            for _, _, ks, status, _, _, _ in self.inflight:
                bad = np.nonzero(np.asarray(status) != 0)[0]
                failed.update(ks[int(r)] for r in bad)
            self.failed = frozenset(failed)
            if getattr(self.ctx, "h", True) is not None:
                for d_P, _, _, _, d_L, _, _ in self.inflight:
                    d_P.free()
                    if d_L is not None:
                        d_L.free()
                for buf in self.release:
                    buf.free()
            self.inflight = self.release = ()
            self.done = True

    def wait_grid(self, k):
        self.wait()
        if k in self.failed:
            raise DensitiesError("no samples in bin")

    def __del__(self):
        try:
            self.wait()
        except Exception:
            pass


def _helper(self):
    """One background thread per lane for C-ABI calls that can run while this thread does host-only scalar work
    (ctypes releases the GIL for the duration of a call).  The thread binds the context's device first."""
    if self._helper_exec is None:
        from concurrent.futures import ThreadPoolExecutor

        self._helper_exec = ThreadPoolExecutor(max_workers=1, thread_name_prefix="gdhip-helper",
                                               initializer=self.ctx.bind_thread)
    return self._helper_exec


def get2DDensities_planned(self, pairs, num_plot_contours=None, get_density=True, _bandwidths=None, meanlikes=False,
                         mask_function=None, **kwargs):
    """One lane of get2DDensities: the whole batched pipeline on this object's context."""
    base_F = kwargs.get("fine_bins_2D", self.fine_bins_2D)
    bco = kwargs.get("boundary_correction_order", self.boundary_correction_order)
    mbc = kwargs.get("mult_bias_correction_order", self.mult_bias_correction_order)
    smooth_scale_2D = float(kwargs.get("smooth_scale_2D", self.smooth_scale_2D))
    if abs(self.max_corr_2D) > 1:
        raise SettingError("max_corr_2D cannot be >=1")
    if bco > 1:
        raise SettingError("unknown boundary_correction_order (expected 0 or 1)")
    ctx = self.ctx
    _hostlog("lane start")
    pa = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    jx, jy = np.ascontiguousarray(pa[:, 0]), np.ascontiguousarray(pa[:, 1])
    flat = pa.ravel()
    used = flat[np.sort(np.unique(flat, return_index=True)[1])].tolist()  # in order of first appearance
    self._init_params(used)
    names = self.paramNames.names
    # the N_eff kernels need nothing but the parameter ranges: they are started first, from the helper thread, and
    # run while the per-pair scalars below are worked out
    neff_f = None
    if (smooth_scale_2D < 0 and _bandwidths is None and not self._timing and not self.use_effective_samples_2D
            and self._lane == 0 and not meanlikes and len(pa) >= 64
            and os.environ.get("GETDIST_AMD_OVERLAP_NEFF", "1") == "1"
            and any(names[j].N_eff_kde is None for j in used)):
        neff_f = self._helper().submit(self._neff_batch, used)
    _hostlog("lane: pairs indexed, N_eff submitted")
    # (binmin, binmax) per parameter: _bin_edges for all parameters at once (they do not depend on the grid size)
    bmin_t, bmax_t = self._bin_edge_arrays(used)
    early_prebin = None
    if (neff_f is not None and base_F == 256 and self.weights is None and len(pa) >= 128
            and hasattr(ctx, "hist2d_prebinned8") and os.environ.get("GETDIST_AMD_U8", "1") == "1"):
        # the byte index columns of the base grid need nothing but these edges: their launch goes out on the second
        # context before the per-pair scalars below are worked out (binning() finds them made)
        twin = self._second_lane()
        self._nlanes = 1
        fw256 = (bmax_t - bmin_t) / 255
        early_prebin = self._lane_thread(twin).submit(twin._index_columns8, {j: (bmin_t[j], fw256[j]) for j in used})
    corrmat = self.getCorrelationMatrix()
    # ---- per-pair scalars (mcsamples.py:1794-1822); bin edges depend on (parameter, F) only
    # vectorised over the pairs: the correlation handling, angle_scale and the grid up-scaling (mcsamples.py:1796-1816)
    actual = np.asarray(corrmat)[jy, jx]
    corr_v = actual.copy()
    full = np.abs(np.abs(corr_v) - 1.0) <= 1e-8
    for q in np.nonzero(full)[0]:
        logging.warning("Parameters are 100%% correlated: %s, %s", names[jx[q]].name, names[jy[q]].name)
    corr_v[full] = np.sign(corr_v[full]) * self.max_corr_2D
    corr_v[np.abs(corr_v) < 0.1] = 0.0
    angle_scale = np.maximum(0.2, np.sqrt(1 - np.minimum(self.max_corr_2D, np.abs(corr_v)) ** 2))
    nbin2D_v = np.round(self.num_bins_2D / angle_scale).astype(np.int64)
    scaled = 192 * (3 / angle_scale).astype(np.int64) // 3
    F_v = np.where((corr_v != 0) & (base_F < scaled) & ((1 / angle_scale).astype(np.int64) > 1), scaled, base_F)
    pj, pj2, pF = jx.tolist(), jy.tolist(), [int(f) for f in F_v.tolist()]
    _hostlog("lane: grid sizes")
    # (fine width, binmin, binmax) per (parameter, F): the table of widths per grid size
    F_list = list(dict.fromkeys(pF))
    fw_t = {F: (bmax_t - bmin_t) / (F - 1) for F in F_list}
    edge_of = {(j, F): (fw_t[F][j], bmin_t[j], bmax_t[j])
               for F in F_list for j in np.unique(np.concatenate([jx[F_v == F], jy[F_v == F]])).tolist()}
    fwx_v = np.empty(len(pa))
    fwy_v = np.empty(len(pa))
    for F in F_list:
        sel_F = F_v == F
        fwx_v[sel_F], fwy_v[sel_F] = fw_t[F][jx[sel_F]], fw_t[F][jy[sel_F]]
    info = []

    def build_info():
        """The per-pair records; nothing here is needed to start the binning."""
        corr_l, actual_l, nbin_l = corr_v.tolist(), actual.tolist(), nbin2D_v.tolist()
        with _Phase(self, "2d.host_pair_scalars"):
            for q, (j, j2, F) in enumerate(zip(pj, pj2, pF)):
                fwx, xbinmin, xbinmax = edge_of[(j, F)]
                fwy, ybinmin, ybinmax = edge_of[(j2, F)]
                info.append(dict(j=j, j2=j2, parx=names[j], pary=names[j2], corr=corr_l[q], actual_corr=actual_l[q],
                                 F=F, nbin2D=nbin_l[q], fwx=fwx, xbinmin=xbinmin, xbinmax=xbinmax, fwy=fwy,
                                 ybinmin=ybinmin, ybinmax=ybinmax))

    _hostlog("lane: bin edges")
    # ---- histograms, one batched launch per grid-size class (pre-binned index columns)
    classes = {F: np.nonzero(F_v == F)[0].tolist() for F in F_list}
    hists, likehists = {}, {}

    def binning(owner=self):
        """prebin + batched 2D histograms on ``owner``'s context (this object, or its second-lane twin)."""
        for F, members in classes.items():
            if (F == 256 and owner.weights is None and not meanlikes and len(members) >= 64
                    and hasattr(owner.ctx, "hist2d_prebinned8") and os.environ.get("GETDIST_AMD_U8", "1") == "1"):
                # the base grid of a unit-weight triangle: byte indices, packed 16-bit counters, one block per pair
                with _Phase(self, "2d.prebin"):
                    wanted = {}
                    for j in dict.fromkeys([pj[k] for k in members] + [pj2[k] for k in members]):
                        fw, bmin, _ = edge_of[(j, 256)]
                        wanted[j] = (bmin, fw)
                    ok = owner._index_columns8(wanted)
                    _hostlog("binning: byte index columns launched")
                if ok:
                    try:
                        with _Phase(self, "2d.hist"):
                            # device addresses per pair from a per-column table (this runs on a helper thread
                            # while the main thread is in Python: as little interpreter work here as possible)
                            table = np.zeros(max(wanted) + 1, dtype=np.uint64)
                            for j in wanted:
                                table[j] = owner._idx_cols[(j, 256, "u8")][0].ptr
                            mem = np.asarray(members, dtype=np.int64)
                            hists[F] = (owner.ctx.hist2d_prebinned8(table[jx[mem]], table[jy[mem]]), members)
                        continue
                    except GdhipError as e:
                        if e.code != -5:  # a 16-bit counter wrapped: the u16 / u32 path below redoes the class
                            raise
            with _Phase(self, "2d.prebin"):
                ix = [owner._index_column(pj[k], F, edge_of[(pj[k], F)][1], edge_of[(pj[k], F)][0]) for k in members]
                iy = [owner._index_column(pj2[k], F, edge_of[(pj2[k], F)][1], edge_of[(pj2[k], F)][0]) for k in members]
            with _Phase(self, "2d.hist"):
                hists[F] = (owner.ctx.hist2d_prebinned(ix, iy, F), members)
                _hostlog("binning: class F=%d done" % F)
                if meanlikes:
                    likehists[F] = self._like_histograms(0, lambda: ctx.hist2d_prebinned(ix, iy, F))

    def plan_args():
        return ([(e["j"], e["j2"]) for e in info], [e["actual_corr"] for e in info],
                [(e["xbinmax"] - e["xbinmin"], e["ybinmax"] - e["ybinmin"]) for e in info], base_F)

    auto_bw = smooth_scale_2D < 0 and _bandwidths is None
    plan = shear = None
    if auto_bw and not self._timing and not self.use_effective_samples_2D:
        # Everything that needs no device result runs on this thread while helper threads sit in the (GIL-free)
        # entry points: the N_eff kernels (exp-bound) on this context, the byte-index binning (LDS-bound) on the
        # second context (own stream and scratch over the same resident samples), then the sheared re-binning
        # (HBM-bound) on this context again; the per-pair records and the branch selection are built meanwhile.
        if neff_f is not None:
            twin = self._second_lane()
            self._nlanes = 1  # only the binning is shared out
            pending = self._lane_thread(twin).submit(binning, twin)
            _hostlog("binning and N_eff submitted")
            try:
                try:
                    build_info()
                    plan, fill_plan = self._bandwidth_plan(*plan_args(), defer_neff=True)
                finally:
                    neff_f.result()
                self._neff_complete(used)  # (multi-rank runs: the other ranks' values, from this thread)
                # a context is not re-entrant: the shear launches start once the N_eff call has returned
                shear_f = self._helper().submit(self._shear_histograms, plan, base_F)
                try:
                    fill_plan()
                finally:
                    shear = shear_f.result()
            finally:
                pending.result()
                if early_prebin is not None:
                    early_prebin.result()
        else:
            self._neff_batch(used)
            pending = self._helper().submit(binning)  # the helper thread is inside this context's entry points
            _hostlog("binning submitted")
            try:
                build_info()
                plan = self._bandwidth_plan(*plan_args())
            finally:
                pending.result()
    else:
        build_info()
        binning()
    # ---- convolution set-up: everything that does not depend on the bandwidths
    _hostlog("binning / N_eff / plan joined")
    npair = len(info)
    # Bookkeeping that no kernel waits for (per-pair records, log messages) is collected here and run once enough
    # convolution batches are queued: between the optimiser's last kernel and the convolution's first one the GPU is
    # idle, so only what decides the window sizes is evaluated there, on arrays.
    deferred = []
    out = [None] * len(info)
    max_bytes = float(os.environ.get("GETDIST_AMD_BATCH_BYTES", 24e9))
    inflight = []  # (device grid buffer, pinned host array, pair indices, status)
    # per-pair flag bits.  Edge masks only on non-periodic axes (mcsamples.py:1688-1703); bits 0/1 = x bot/top,
    # 2/3 = y bot/top, 4/5 = x/y periodic.
    nmax = max(used) + 1
    lim_bits, per_bit, has_lim = np.zeros(nmax, np.int64), np.zeros(nmax, np.int64), np.zeros(nmax, bool)
    for j in used:
        p_ = names[j]
        lim_bits[j] = 0 if p_.periodic else (1 if p_.has_limits_bot else 0) | (2 if p_.has_limits_top else 0)
        per_bit[j] = 1 if p_.periodic else 0
        has_lim[j] = bool(p_.has_limits)
    has_prior_v = has_lim[jx] | has_lim[jy] | (mask_function is not None)  # mcsamples.py:1794
    flags_v = lim_bits[jx] | (per_bit[jx] << 4) | (lim_bits[jy] << 2) | (per_bit[jy] << 5) | (has_prior_v.astype(np.int64) << 6)
    group_v = (flags_v & 48) * 2 + (has_prior_v & (bco >= 0))
    # window scales in fine-grid units, window half-widths: filled as the bandwidths become known
    rx_v, ry_v, cc_v = np.full(npair, np.nan), np.full(npair, np.nan), np.full(npair, np.nan)
    smooth_v, winw_v = np.full(npair, np.nan), np.zeros(npair, dtype=np.int64)
    wv = np.full((npair, 3), np.nan)

    def set_scales(ks, rx_k, ry_k, cc_k):
        rx_v[ks], ry_v[ks], cc_v[ks] = rx_k, ry_k, cc_k
        smooth_v[ks] = np.maximum(rx_v[ks], ry_v[ks])
        winw_v[ks] = np.maximum(1, np.rint(2.5 * smooth_v[ks]).astype(np.int64))  # max(1, int(round(2.5 * smooth_scale)))

    def warn_coarse():
        for k in np.nonzero(smooth_v < 2)[0].tolist():
            logging.warning("fine_bins_2D not large enough for optimal density: %s, %s", info[k]["parx"].name,
                            info[k]["pary"].name)

    deferred.append(warn_coarse)

    enqueue_only = (hasattr(ctx, "density2d_enqueue") and not self._timing and not meanlikes and mask_function is None
                    and os.environ.get("GETDIST_AMD_ASYNC_CONVOLVE", "1") == "1")
    release = []
    status_all, status_at = (ctx.pinned_array((npair,), np.int32) if enqueue_only else None), [0]
    # With every batch enqueued without waiting, the call may return while the last result copies are still in
    # flight: the grids' first reader (or the next batched call, or the collection of the results) completes them.
    lazy = (enqueue_only and get_density and hasattr(ctx, "copy_mark")
            and os.environ.get("GETDIST_AMD_LAZY_RESULTS", "1") == "1")
    conv_ctxs, batch_no, main_no, widths_complete = [ctx], [0], [0], [True]
    # Second stream for the convolution.  A small call (one rank's share of a triangle) cannot fill the chip with one
    # batch's kernels: its batches alternate between the streams of the two contexts.  A large call keeps its main
    # grid class on this context and sends the few pairs of the up-scaled classes (large frames, a handful of grids
    # per launch) to the second one, where they run beside the main class instead of after it.
    side_classes = set()
    if lazy and self._lane == 0 and not self._timing and self._context_factory is not None and npair >= self.CONV_TWO_STREAMS_PAIRS[0]:
        if npair > self.CONV_TWO_STREAMS_PAIRS[1]:
            side_classes = {F_ for F_, (_, mem_) in hists.items() if len(mem_) < 64}
            if len(side_classes) == len(hists):
                side_classes = set()
        nlanes = self._nlanes
        conv_ctxs.append(self._second_lane().ctx)  # idle by now: its binning has been joined
        self._nlanes = nlanes
    completion = _PendingResults(ctx, inflight, release, conv_ctxs[1:]) if lazy else None  # shares the two lists filled below
    if lazy and self._pending_results is not None:
        # the previous call's copies landed long ago: its device blocks are handed back while this call's first
        # batches compute, not between the last enqueue and the caller's next launch
        deferred.append(self._pending_results.wait)
    parked, self._parked = [getattr(self, "_parked", None)], None
    deferred.append(parked.clear)  # the previous call's per-pair records (see the end of this function)
    import functools

    # the grid axes of every (parameter, F) in use, all at once: np.linspace(lo, hi, F) written out
    # (k * step + start, last point = stop) on one 2D array per F
    ax_cache = {}

    def make_axes():
        for F_ in F_list:
            js_ = np.unique(np.concatenate([jx[F_v == F_], jy[F_v == F_]])).tolist()
            lo_ = np.array([edge_of[(j, F_)][1] for j in js_], dtype=np.float64)
            hi_ = np.array([edge_of[(j, F_)][2] for j in js_], dtype=np.float64)
            A = np.arange(F_, dtype=np.float64)[None, :] * ((hi_ - lo_) / (F_ - 1))[:, None] + lo_[:, None]
            A[:, -1] = hi_
            for row, j in enumerate(js_):
                ax_cache[(j, F_)] = (A[row], A[row, 1] - A[row, 0], (names[j].range_min, names[j].range_max))

    deferred.append(make_axes)  # needed by the result objects only
    assembled = [0]
    sync_state = [False]

    def assemble_new():
        """Result objects of the batches enqueued since the last call: they only hold views of the page-locked
        arrays, so they are built while those batches compute and copy -- batch by batch, not after the last
        enqueue, where the host work would sit between this call's kernels and the caller's next ones."""
        for d_P, P, ks, status, d_L, L, levels in inflight[assembled[0]:]:
            F = P.shape[1]
            lev_state = None if levels is None else np.asarray(levels[1]).tolist()
            if lev_state is not None and -5 in lev_state and not sync_state[0]:  # a grid left to the host reads P
                for c in conv_ctxs:
                    c.copy_sync()
                sync_state[0] = True
            ncont = None
            for row, k in enumerate(ks):
                e = info[k]
                ax, sx, vrx = ax_cache[(e["j"], F)]
                ay, sy, vry = ax_cache[(e["j2"], F)]
                contours = None
                if lev_state is not None:
                    if lev_state[row] == 0:
                        contours = levels[0][row].copy()
                    elif lev_state[row] == -4:
                        raise DensitiesError("Contour level outside plotted ranges")
                    else:
                        ncont = levels[0].shape[1]
                dens = Density2D._from_fields(dict(
                    x=ax, y=ay, axes=[ay, ax], spacing=sx * sy, view_ranges=[vrx, vry], mask=e.get("mask"),
                    likes=None if L is None else L[row], contours=contours, spl=None, _P=P[row],
                    _wait=functools.partial(completion.wait_grid, k) if lazy else None,
                    bandwidth=e.get("bandwidth"), bandwidth_branch=e.get("branch"), kopt=e.get("kopt")))
                if contours is None and lev_state is not None:
                    # more exactly equal grid values at the level than the kernel's tie list holds
                    dens.contours = dens.getContourLevels(self.contours[:ncont])
                out[k] = dens
        assembled[0] = len(inflight)

    def run_deferred():
        while deferred:
            deferred.pop(0)()

    def run_class(F, d_hist, members, only=None, force_ctx=None):
        """Convolve the pairs of one grid-size class (``only``: a mask over the pair indices -- the members outside
        it are left for a later call; ``force_ctx``: the context whose stream takes the batches): a generator that
        returns control after every batch it has enqueued, so that the caller can interleave the classes' batches."""
        mem = np.asarray(members, dtype=np.int64)
        pos_all = np.arange(len(mem)) if only is None else np.nonzero(only[mem])[0]
        if not len(pos_all):
            return
        # batches of a few hundred grids (default cap 320) keep the FFTs efficient and the D2H copy of batch k hidden behind the
        # convolution of batch k+1; only the last (small) batch's copy is exposed at the end
        max_batch = max(1, min(int(max_bytes // (F * F * 8 * 30)), int(os.environ.get("GETDIST_AMD_MAX_BATCH", 320))))
        batches = []
        first_batch = int(os.environ.get("GETDIST_AMD_FIRST_BATCH", 128))
        gk = np.full(len(mem), -1, dtype=np.int64)
        gk[pos_all] = group_v[mem[pos_all]]
        frame = {w_: next_fft_size(F + 2 * w_) for w_ in np.unique(winw_v[mem[pos_all]]).tolist()}
        S_v = np.zeros(len(mem), dtype=np.int64)
        S_v[pos_all] = [frame[w_] for w_ in winw_v[mem[pos_all]].tolist()]
        for g_ in dict.fromkeys(gk[pos_all].tolist()):  # groups in order of first appearance
            in_g = np.nonzero(gk == g_)[0]
            # sub-batches of equal frame size S >= F + 2 winw.  A pair's frame follows from its own window only (no
            # merging of small sub-batches into the next size): its grid is then the same bit for bit in whatever
            # call, share or chunk it is computed
            for S in np.unique(S_v[in_g]).tolist():
                pos_S = in_g[S_v[in_g] == S]
                cur = [(int(pos), members[pos]) for pos in pos_S.tolist()]
                # a short first batch starts the result copies early; from then on a batch's copy (PCIe) is shorter
                # than the next batch's kernels, so only the last batch's copy is exposed
                s0 = 0
                if not batches and len(cur) > first_batch:
                    batches.append(cur[:first_batch])
                    s0 = first_batch
                for s1 in range(s0, len(cur), max_batch):
                    batches.append(cur[s1:s1 + max_batch])
        if mask_function is not None:
            batches = [[item] for b in batches for item in b]  # the callback edits one pair's mask at a time
        for sel in batches:
            if mask_function is not None:
                (pos, k), = sel
                e = info[k]
                w_ = int(winw_v[k])
                prior_mask = np.ones((F + 2 * w_, F + 2 * w_))
                mask_function(e["xbinmin"] - w_ * e["fwx"], e["ybinmin"] - w_ * e["fwy"], e["fwx"], e["fwy"], prior_mask)
                e["mask"] = bool_mask = prior_mask[w_:-w_, w_:-w_] < 1e-8
                mask_bc = mask_mbc = None
                if bco >= 0:
                    _set_edge_mask_2d(e["parx"], e["pary"], prior_mask, w_)
                    mask_bc = prior_mask.copy()
                if mbc:
                    _set_all_edge_mask_2d(prior_mask, w_, e["parx"].periodic, e["pary"].periodic)
                    mask_mbc = prior_mask
                run_deferred()
                with _Phase(self, "2d.convolve"):
                    d_P, status = ctx.density2d_masked(d_hist, pos, F, float(rx_v[k]), float(ry_v[k]), float(cc_v[k]), w_,
                                                       int(flags_v[k]), bco, mbc, mask_bc, mask_mbc, bool_mask)
                levels = None
                if not get_density:
                    ncontours = len(self.contours)
                    if num_plot_contours:
                        ncontours = min(num_plot_contours, ncontours)
                    levels = ctx.contour_levels(d_P, 1, F, self.contours[:ncontours])
                d_L = L = None
                if meanlikes:
                    # the mean-likelihood grid does not see the mask (mcsamples.py:1886-1903 precede it): the pair's
                    # ordinary likes2d call
                    d_one, d_lone = ctx.alloc(F * F * 8), ctx.alloc(F * F * 8)
                    self._gather_device(d_hist, d_one, [pos], F * F * 8)
                    self._gather_device(likehists[F], d_lone, [pos], F * F * 8)
                    ka1 = np.asarray([k], dtype=np.int64)
                    d_L, lstatus = ctx.likes2d(d_one, d_lone, 1, F, rx_v[ka1], ry_v[ka1], cc_v[ka1], winw_v[ka1], flags_v[ka1], mbc)
                    d_one.free()
                    d_lone.free()
                    if np.any(lstatus != 0):
                        raise DensitiesError("no likelihood weight in any bin")
                    L = d_L.to_host_async((1, F, F))
                inflight.append((d_P, d_P.to_host_async((1, F, F)), [k], status, d_L, L, levels))
                assemble_new()
                yield
                continue
            # a small call (one rank's share of a triangle, a handful of pairs) cannot fill the chip with one
            # batch's kernels: its batches go alternately to the streams of the two contexts and run side by side
            if force_ctx is not None:
                bctx = force_ctx
            elif side_classes or npair > self.CONV_TWO_STREAMS_PAIRS[1]:
                bctx = conv_ctxs[1 if F in side_classes else 0]
            else:
                bctx = conv_ctxs[batch_no[0] % len(conv_ctxs)]
            batch_no[0] += 1
            if widths_complete[0] and (bctx is ctx or len(sel) >= 64):
                main_no[0] += 1
            if [pos for pos, _ in sel] == list(range(len(members))):
                d_sub, own = d_hist, False
            else:
                d_sub, own = bctx.alloc(len(sel) * F * F * 8), True
                self._gather_device(d_hist, d_sub, [pos for pos, _ in sel], F * F * 8, ctx=bctx)
            ks = [k for _, k in sel]
            ka = np.asarray(ks, dtype=np.int64)
            with _Phase(self, "2d.convolve"):
                if enqueue_only:
                    # returns once enqueued: the next batch is prepared, and at the end the result objects are
                    # built, while this one computes; its status words land in page-locked memory
                    status = status_all[status_at[0]:status_at[0] + len(sel)]
                    status_at[0] += len(sel)
                    d_P = bctx.density2d_enqueue(d_sub, len(sel), F, rx_v[ka], ry_v[ka], cc_v[ka], winw_v[ka], flags_v[ka],
                                                 bco, mbc, status)
                else:
                    d_P, status = ctx.density2d(d_sub, len(sel), F, rx_v[ka], ry_v[ka], cc_v[ka], winw_v[ka], flags_v[ka],
                                                bco, mbc)
            # the first batch is short (it starts the result copies early) and would be through before the
            # bookkeeping: that runs once every bandwidth is known and two more batches are queued
            if main_no[0] >= 2:
                run_deferred()
            levels = None
            if not get_density:  # contour levels on the device while the grids are still resident (densities.py:19-56)
                ncontours = len(self.contours)
                if num_plot_contours:
                    ncontours = min(num_plot_contours, ncontours)
                levels = bctx.contour_levels(d_P, len(sel), F, self.contours[:ncontours])
            d_L = L = None
            if meanlikes:
                if own:
                    d_lsub = ctx.alloc(len(sel) * F * F * 8)
                    self._gather_device(likehists[F], d_lsub, [pos for pos, _ in sel], F * F * 8)
                else:
                    d_lsub = likehists[F]
                d_L, lstatus = ctx.likes2d(d_sub, d_lsub, len(sel), F, rx_v[ka], ry_v[ka], cc_v[ka], winw_v[ka], flags_v[ka], mbc)
                if own:
                    d_lsub.free()
                if np.any(lstatus != 0):
                    raise DensitiesError("no likelihood weight in any bin")
                L = d_L.to_host_async((len(sel), F, F))
            if own:
                release.append(d_sub)  # freeing waits for the stream: after the last batch
            # the copy runs on the copy stream while the next batch computes
            inflight.append((d_P, d_P.to_host_async((len(sel), F, F)), ks, status, d_L, L, levels))
            if not self._timing and not deferred:
                assemble_new()
            yield

    # largest class (in bytes) first: the copy of the last, smallest one is the only exposed one
    order = sorted(hists.items(), key=lambda kv: -len(kv[1][1]) * kv[0] * kv[0])

    def enqueue_all():
        """Every class, all pairs: classes that go to the second stream are queued there right after the main class's
        first batch."""
        main = [run_class(F, d_hist, members) for F, (d_hist, members) in order if F not in side_classes]
        side = [run_class(F, d_hist, members) for F, (d_hist, members) in order if F in side_classes]
        if side and main:
            next(main[0], None)  # the main class's short first batch goes out before the second stream is fed
        for gen in side + main:
            for _ in gen:
                pass

    # ---- bandwidths: the whole optimiser (fixed point, functionals, TNC) runs on the device
    all_k = np.arange(npair)
    if smooth_scale_2D < 0:

        def book_widths():  # (queued before anything is enqueued: the result objects carry the triples)
            for e, bw_k in zip(info, wv.tolist()):
                e["bandwidth"] = tuple(bw_k)

        deferred.append(book_widths)
        if _bandwidths is not None:
            wv[:] = np.array(list(_bandwidths), dtype=np.float64).reshape(npair, 3)
            set_scales(all_k, wv[:, 0] * abs(smooth_scale_2D) / fwx_v, wv[:, 1] * abs(smooth_scale_2D) / fwy_v, wv[:, 2])
            enqueue_all()
        else:
            if plan is None:
                with _Phase(self, "2d.host_bandwidth_plan"):
                    plan = self._bandwidth_plan(*plan_args())
            # A large call on two streams: the optimiser's launch of the base grid is cut in two, and the first part's
            # pairs are convolved on the second context's stream while the second part is still being optimised on
            # this one (the optimiser re-streams its matrices from the memory-side cache, the convolution is
            # arithmetic in LDS: they share the chip well).
            pipelined = len(conv_ctxs) > 1 and npair > self.CONV_TWO_STREAMS_PAIRS[1]

            def book_plan(plan=plan):
                for e, pl in zip(info, plan):
                    e["branch"], e["kopt"] = pl["branch"], pl["kopt"]

            enqueueing = []  # the first part's enqueue, running on the second context's thread

            def enqueue_part(only, last):
                for F, (d_hist, members) in order:
                    target = conv_ctxs[1] if (F in side_classes or not last) else conv_ctxs[0]
                    for _ in run_class(F, d_hist, members, only=only, force_ctx=target):
                        pass

            def on_chunk(ks, last, W):
                wv[ks] = W[ks]
                set_scales(ks, wv[ks, 0] * abs(smooth_scale_2D) / fwx_v[ks], wv[ks, 1] * abs(smooth_scale_2D) / fwy_v[ks],
                           wv[ks, 2])
                only = np.zeros(npair, dtype=bool)
                only[ks] = True
                if not last:
                    # the second context's own thread enqueues this part's convolution while this thread goes
                    # straight on to the next optimiser launch (a blocking call that releases the interpreter lock)
                    enqueueing.append(self._lane_thread(self._second_lane()).submit(enqueue_part, only, False))
                    return
                for f in enqueueing:
                    f.result()
                widths_complete[0] = True
                _hostlog("bandwidths done")
                enqueue_part(only, True)

            with _Phase(self, "2d.bandwidth.device"):
                if pipelined:
                    widths_complete[0] = False
                    try:
                        self._bandwidth_2d(plan, hists, pF, base_F, mbc, shear=shear, deferred=deferred, on_chunk=on_chunk,
                                           first_fraction=self.KOPT_FIRST_FRACTION, more_deferred=book_plan)
                    except BaseException:
                        # a later optimiser launch failed ("bias not positive definite", a device error) while the
                        # second context's thread may still be enqueueing an earlier part: let it finish and the
                        # queued work drain before the error leaves -- the contexts are not re-entrant, and the
                        # blocks of this call are released only once nothing runs on them
                        for f in enqueueing:
                            try:
                                f.result()
                            except BaseException:
                                pass
                        if lazy:
                            try:
                                completion.mark()
                                completion.wait()
                            except BaseException:
                                pass
                        raise
                else:
                    W = self._bandwidth_2d(plan, hists, pF, base_F, mbc, shear=shear, deferred=deferred,
                                           more_deferred=book_plan)
                    wv[:] = W
                    set_scales(all_k, wv[:, 0] * abs(smooth_scale_2D) / fwx_v, wv[:, 1] * abs(smooth_scale_2D) / fwy_v, wv[:, 2])
                    _hostlog("bandwidths done")
                    enqueue_all()

    else:
        if smooth_scale_2D < 1.0:
            set_scales(all_k, smooth_scale_2D * np.array([e["parx"].err for e in info]) / fwx_v,
                       smooth_scale_2D * np.array([e["pary"].err for e in info]) / fwy_v,
                       np.array([e["corr"] for e in info], dtype=np.float64))
        else:
            fixed = np.array([smooth_scale_2D * e["F"] / e["nbin2D"] for e in info], dtype=np.float64)
            set_scales(all_k, fixed, fixed, np.array([e["corr"] for e in info], dtype=np.float64))
        enqueue_all()
    _ph_asm = _Phase(self, "2d.host_assemble_results")
    _hostlog("classes enqueued (%s)" % ", ".join("F=%d: %d pairs" % (F, len(m_)) for F, (_, m_) in order))
    run_deferred()  # (nothing was enqueued: no pairs)
    release += [d_hist for d_hist, _ in hists.values()] + list(likehists.values())
    _hostlog("all batches enqueued")
    _ph_asm.__enter__()
    assemble_new()
    synced = sync_state[0]
    if lazy:
        completion.mark()  # after the last copy of this call
    _ph_asm.__exit__()
    _hostlog("results assembled")
    if lazy:
        previous, self._pending_results = self._pending_results, completion
        if previous is not None:
            # the previous call's copies are ahead of this call's on the copy stream: completing it here costs no
            # waiting, and its device blocks are released even if nobody ever read its grids
            previous.wait()
        # The per-pair records of a triangle are a few thousand small objects: tearing them down on return would
        # sit between this call's last enqueue and the caller's next launch.  They are parked and dropped by the
        # next batched call while its first batches compute (or with this object).
        self._parked = (info, plan)
        return out
    if not synced:
        with _Phase(self, "2d.d2h_wait"):
            for c in conv_ctxs:
                c.copy_sync()
    failed = any(np.any(np.asarray(status) != 0) for _, _, _, status, _, _, _ in inflight)
    for d_P, P, ks, status, d_L, L, levels in inflight:
        d_P.free()
        if d_L is not None:
            d_L.free()
    for buf in release:
        buf.free()
    if failed:
        raise DensitiesError("no samples in bin")
    return out


def get1DDensities_planned(self, js, pars, fine_bins, num_bins, smooth_scale_1D, bco, mbc, meanlikes, kwargs):
    """The Python-planned sequence of get1DDensities (gd_hist1d, gd_isj1d, the scalar tail of getAutoBandwidth1D in
    Python, gd_density1d): the comparison of the native gd_density1d_batch route (mcsamples.py:1500-1686 of the reference)."""
    edges = []
    for par in pars:
        if par.range_max - par.range_min <= 0:
            raise MCSamplesError("Parameter range is <= 0: " + par.name)
        edges.append(self._bin_edges(par, fine_bins))
    hist = self.ctx.hist1d(js, [e[1] for e in edges], [e[0] for e in edges], fine_bins)
    smooth, winw, flags = [], [], []
    isj_h = isj_status = None
    if smooth_scale_1D <= 0:
        self._neff_batch(js)
        isj_h, isj_status = self.ctx.isj1d(hist, [self._get1DNeff(par, j) for j, par in zip(js, pars)])
    for b, (j, par) in enumerate(zip(js, pars)):
        fine_width, binmin, binmax = edges[b]
        paramrange = par.range_max - par.range_min
        width = paramrange / (num_bins - 1)
        if smooth_scale_1D <= 0:
            N_eff = self._get1DNeff(par, j)
            bandwidth = self._bandwidth_1d(None if isj_status[b] else isj_h[b], par, N_eff, mbc, bco) * (binmax - binmin)
            bandwidth = min(bandwidth, paramrange / 4)
            smooth_1D = bandwidth * abs(smooth_scale_1D) / fine_width
        elif smooth_scale_1D < 1.0:
            smooth_1D = smooth_scale_1D * par.err / fine_width
        else:
            smooth_1D = smooth_scale_1D * width / fine_width
        if smooth_1D < 2:
            logging.warning("fine_bins not large enough to well sample smoothing scale - " + par.name)
        smooth_1D = min(max(1.0, smooth_1D), fine_bins // 2)
        smooth.append(smooth_1D)
        winw.append(min(int(round(2.5 * smooth_1D)), ((fine_bins - 1) if par.periodic else fine_bins) // 2 - 2))
        flags.append((1 if par.has_limits_bot else 0) | (2 if par.has_limits_top else 0) | (4 if par.periodic else 0))
    P, status = self.ctx.density1d(hist, smooth, winw, flags, bco, mbc)
    if np.any(status != 0):
        raise DensitiesError("no samples in bin")
    return self._finish_1d(js, pars, edges, P, hist, smooth, winw, flags, fine_bins, meanlikes, kwargs)


def install():
    _M.MCSamples._planned_route = get2DDensities_planned
    _M.MCSamples._planned_route_1d = get1DDensities_planned
    _M.MCSamples._helper = _helper
