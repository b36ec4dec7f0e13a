"""
TEST INFRASTRUCTURE: drives getdist_amd/csrc/batch2d.hpp -- the plan and the choreography of gd_density2d_batch,
compiled for the host by tests/native/build.py -- with the table of device entry points bound to the numpy context
double (tests/fake_ctx.py).  `HarnessContext` is a FakeContext with the four batch methods of getdist_amd._lib.Context,
so `MCSamples(..., _context_factory=HarnessContext).get2DDensities(pairs)` takes the product's native route on the CPU
and its grids can be compared with the Python-planned route on the same double.
"""

import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "native"))

from fake_ctx import FakeBuf, FakeContext  # noqa: E402

_p, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_pd, _pi32, _pi64, _pp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_void_p)

# field order = struct gdb::Ops (batch2d.hpp)
_OPS = [
    ("bind_thread", C.CFUNCTYPE(C.c_int, _p)),
    ("num_rows", C.CFUNCTYPE(C.c_int, _p, _pi64, _pi64)),
    ("weights_kind", C.CFUNCTYPE(C.c_int, _p, _pi32)),
    ("dev_alloc", C.CFUNCTYPE(C.c_int, _p, _i64, _pp)),
    ("dev_free", C.CFUNCTYPE(C.c_int, _p, _p)),
    ("prebin8_batch", C.CFUNCTYPE(C.c_int, _p, _pi32, _i32, _pd, _pd, _i32, _pp, _pi64)),
    ("hist2d_prebinned8", C.CFUNCTYPE(C.c_int, _p, _i32, _pp, _pp, _p)),
    ("prebin8_hist2d", C.CFUNCTYPE(C.c_int, _p, _pi32, _i32, _pd, _pd, _pp, _pi64, _i32, _pp, _pp, _p)),
    ("prebin", C.CFUNCTYPE(C.c_int, _p, _i32, _f64, _f64, _i32, _p)),
    ("hist2d_prebinned", C.CFUNCTYPE(C.c_int, _p, _i32, _pp, _pp, _i32, _p)),
    ("minmax_affine", C.CFUNCTYPE(C.c_int, _p, _i32, _pi32, _pi32, _pd, _pd, _pd)),
    ("hist2d_sheared", C.CFUNCTYPE(C.c_int, _p, _i32, _pi32, _pi32, _pd, _pd, _pd, _pd, _pd, _pd, _i32, _p)),
    ("kopt2d", C.CFUNCTYPE(C.c_int, _p, _i32, _i32, _p, _pd, _pi32, _pd, _pd, _pd)),
    ("gather_items", C.CFUNCTYPE(C.c_int, _p, _p, _i64, _p, _pi32, _i32, _i64)),
    ("density2d_enqueue", C.CFUNCTYPE(C.c_int, _p, _i32, _i32, _p, _pi32, _pd, _pd, _pd, _pi32, _pi32, _i32, _i32, _p, _pi32)),
    ("d2h_async", C.CFUNCTYPE(C.c_int, _p, _p, _p, _i64)),
    ("copy_mark", C.CFUNCTYPE(C.c_int, _p, _pi32)),
    ("copy_wait", C.CFUNCTYPE(C.c_int, _p, _i32)),
    ("copy_sync", C.CFUNCTYPE(C.c_int, _p)),
    ("contour_levels", C.CFUNCTYPE(C.c_int, _p, _i32, _i32, _p, _pd, _i32, _pd, _pi32)),
    ("autocov_lags_batch", C.CFUNCTYPE(C.c_int, _p, _pi32, _i32, _pd, _i64, _i32, _pd)),
    ("kde_lag_sums_batch", C.CFUNCTYPE(C.c_int, _p, _pi32, _i32, _pd, _pi64, _i32, _pd)),
    ("kde_lag_sums", C.CFUNCTYPE(C.c_int, _p, _i32, _f64, _pi64, _i32, _pd)),
    ("last_error", C.CFUNCTYPE(C.c_char_p, _p)),
    ("create_aux", C.CFUNCTYPE(C.c_int, _p, _pp)),
    ("destroy_aux", C.CFUNCTYPE(C.c_int, _p)),
    ("kopt2d_enqueue", C.CFUNCTYPE(C.c_int, _p, _i32, _i32, _p, _pd, _pi32, _pd, _pd, _p, _pi32)),
    ("kopt2d_finish", C.CFUNCTYPE(C.c_int, _p, _p, _i32, _i32, _p, _pd)),
    ("comm_world", C.CFUNCTYPE(C.c_int, _p)),
    ("comm_allreduce_sum", C.CFUNCTYPE(C.c_int, _p, _pd, _i64)),
    ("stream_priority", C.CFUNCTYPE(C.c_int, _p, C.c_int)),
    ("prebin_batch", C.CFUNCTYPE(C.c_int, _p, _pi32, _i32, _pd, _pd, _i32, _pp)),
]


class Ops(C.Structure):
    _fields_ = _OPS


_CTX = {}   # handle -> HarnessContext
_BUF = {}   # handle -> FakeBuf
_next = [0x1000]
_KOPT_CACHE = {}
CALLS = []  # (op, details) log for tests that look at the choreography


def _arr(ptr, n, dtype=np.float64):
    return np.ctypeslib.as_array(ptr, shape=(n,)) if n else np.zeros(0, dtype=dtype)


def _guard(fn):
    def wrapped(*a):
        try:
            return fn(*a)
        except Exception:  # never unwind through the C frames; the harness reports a device failure
            import traceback

            traceback.print_exc()
            return -3

    return wrapped


def _make_ops():
    def ctx_of(h):
        return _CTX[int(h)]

    def buf_of(p):
        return _BUF[int(p)]

    def buf_at(p):  # (block, byte offset) of a pointer INTO a block (the second binning launch fills the rows behind the first's)
        p = int(p)
        if p in _BUF:
            return _BUF[p], 0
        base = max(b for b in list(_BUF) if b <= p)
        assert p - base < _BUF[base].nbytes, "a pointer outside every block"
        return _BUF[base], p - base

    def bind_thread(h):
        return 0

    def num_rows(h, N, n):
        c = ctx_of(h)
        N[0], n[0] = c.N, c.n
        return 0

    def weights_kind(h, out):
        out[0] = 0 if ctx_of(h).w is None else 1
        return 0

    def dev_alloc(h, nbytes, out):
        handle = _next[0]
        _next[0] += (int(nbytes) // 0x1000 + 2) * 0x1000  # handles are addresses: pointers into a block must not collide
        _BUF[handle] = FakeBuf(None, int(nbytes))
        out[0] = handle
        return 0

    def dev_free(h, p):
        _BUF.pop(int(p), None)
        return 0

    def prebin8_batch(h, cols, ncols, binmin, width, F, d_idx, bad):
        c = ctx_of(h)
        CALLS.append(("prebin8_batch", c.lane, ncols))
        for q in range(ncols):
            ix = ((c.s[:, cols[q]] - binmin[q]) / width[q] + 0.5).astype(np.int64)
            bad[q] = int(np.sum((ix < 0) | (ix >= F)))
            buf_of(d_idx[q]).a = ix
        return 0

    def hist2d_prebinned8(h, B, ix, iy, d_hist):
        c = ctx_of(h)
        CALLS.append(("hist2d_prebinned8", c.lane, B))
        w = c._w()
        H = np.array([np.bincount(buf_of(ix[q]).a + buf_of(iy[q]).a * 256, weights=w, minlength=65536).reshape(256, 256)
                      for q in range(B)])
        if H.max() > 65535:
            return -5  # a 16-bit counter would have wrapped
        dst, off = buf_at(d_hist)
        if off == 0:
            dst.a = H
        else:
            first = off // (65536 * 8)
            dst.a = np.concatenate([np.asarray(dst.a, dtype=np.float64).reshape(-1, 256, 256)[:first], H])
        return 0

    def prebin8_hist2d(h, cols, ncols, binmin, width, d_idx, bad, B, ix, iy, d_hist):
        if ncols:
            prebin8_batch(h, cols, ncols, binmin, width, 256, d_idx, bad)
            if any(bad[q] for q in range(ncols)):
                return -5
        return hist2d_prebinned8(h, B, ix, iy, d_hist)

    def prebin(h, col, binmin, width, F, d_idx):
        c = ctx_of(h)
        buf_of(d_idx).a = c.prebin(col, binmin, width, F).a
        return 0

    def prebin_batch(h, cols, ncols, binmin, width, F, d_idx):
        c = ctx_of(h)
        CALLS.append(("prebin_batch", c.lane, ncols, F))
        for q in range(ncols):
            buf_of(d_idx[q]).a = c.prebin(cols[q], binmin[q], width[q], F).a
        return 0

    def hist2d_prebinned(h, B, ix, iy, F, d_hist):
        c = ctx_of(h)
        CALLS.append(("hist2d_prebinned", c.lane, B, F))
        buf_of(d_hist).a = c.hist2d_prebinned([buf_of(ix[q]) for q in range(B)], [buf_of(iy[q]) for q in range(B)], F).a
        return 0

    def minmax_affine(h, B, ci, cj, a, b, out):
        c = ctx_of(h)
        CALLS.append(("minmax_affine", c.lane, B))
        res = c.minmax_affine([ci[q] for q in range(B)], [cj[q] for q in range(B)], [a[q] for q in range(B)],
                              [b[q] for q in range(B)])
        _arr(out, 2 * B)[:] = np.asarray(res).ravel()
        return 0

    def hist2d_sheared(h, B, ci, cj, r0, r1, xmin, dx, ymin, dy, F, d_hist):
        c = ctx_of(h)
        CALLS.append(("hist2d_sheared", c.lane, B))
        lst = lambda p: [p[q] for q in range(B)]  # noqa: E731
        buf_of(d_hist).a = c.hist2d_sheared(lst(ci), lst(cj), lst(r0), lst(r1), lst(xmin), lst(dx), lst(ymin), lst(dy), F).a
        return 0

    def kopt2d(h, B, F, d_hist, neff, do_corr, fallback_t, corr, out):
        c = ctx_of(h)
        CALLS.append(("kopt2d", c.lane, B, F))
        H = np.asarray(buf_of(d_hist).a, dtype=np.float64).reshape(-1, F * F)[:B]
        ne, dc, fb, co = _arr(neff, B).copy(), _arr(do_corr, B, np.int32).copy(), _arr(fallback_t, B).copy(), _arr(corr, B).copy()
        o = _arr(out, 12 * B).reshape(B, 12)
        for q in range(B):
            res = c.kopt2d_cached(H[q].reshape(F, F), ne[q], dc[q], fb[q], co[q])
            o[q] = res
        return 0

    def gather_items(h, d_dst, dst_first, d_src, index, count, item_bytes):
        c = ctx_of(h)
        CALLS.append(("gather_items", c.lane, count))
        src = np.asarray(buf_of(d_src).a, dtype=np.float64).reshape(-1, item_bytes // 8)
        picked = src[[index[q] for q in range(count)]]
        dst = buf_of(d_dst)
        if dst_first == 0:
            dst.a = picked.copy()
        else:
            dst.a = np.concatenate([np.asarray(dst.a).reshape(-1, item_bytes // 8)[:dst_first], picked])
        return 0

    def density2d_enqueue(h, B, F, d_hist, hist_index, rx, ry, corr, winw, flags, bco, mbc, d_P, status):
        c = ctx_of(h)
        CALLS.append(("density2d_enqueue", c.lane, B, F))
        src = np.asarray(buf_of(d_hist).a, dtype=np.float64).reshape(-1, F, F)
        view = FakeBuf(src[[hist_index[q] for q in range(B)]] if hist_index else src[:B])
        st = np.zeros(B, dtype=np.int32)
        out = c.density2d_enqueue(view, B, F, _arr(rx, B).copy(), _arr(ry, B).copy(), _arr(corr, B).copy(),
                                  _arr(winw, B, np.int32).copy(), _arr(flags, B, np.int32).copy(), bco, mbc, st)
        _arr(status, B, np.int32)[:] = st
        buf_of(d_P).a = out.a
        return 0

    def d2h_async(h, dst, d_src, nbytes):
        a = np.asarray(buf_of(d_src).a, dtype=np.float64).ravel()[:nbytes // 8]
        C.memmove(dst, a.ctypes.data, nbytes)
        return 0

    def copy_mark(h, tok):
        tok[0] = 0
        return 0

    def copy_wait(h, tok):
        return 0

    def copy_sync(h):
        return 0

    def contour_levels(h, B, F, d_P, contours, nc, out, status):
        c = ctx_of(h)
        view = FakeBuf(np.asarray(buf_of(d_P).a, dtype=np.float64).reshape(-1, F, F)[:B])
        lv, st = c.contour_levels(view, B, F, _arr(contours, nc).copy())
        _arr(out, B * nc)[:] = np.asarray(lv).ravel()
        _arr(status, B, np.int32)[:] = st
        return 0

    def autocov_lags_batch(h, cols, ncols, means, k0, nlags, out):
        c = ctx_of(h)
        CALLS.append(("autocov_lags_batch", c.lane, ncols))
        _arr(out, ncols * nlags)[:] = c.autocov_lags_batch([cols[q] for q in range(ncols)], [means[q] for q in range(ncols)],
                                                           k0, nlags).ravel()
        return 0

    def kde_lag_sums_batch(h, cols, ncols, inv4s2, lags, nlags, out):
        c = ctx_of(h)
        CALLS.append(("kde_lag_sums_batch", c.lane, ncols))
        _arr(out, ncols * nlags)[:] = c.kde_lag_sums_batch([cols[q] for q in range(ncols)], [inv4s2[q] for q in range(ncols)],
                                                           [lags[q] for q in range(nlags)]).ravel()
        return 0

    def kde_lag_sums(h, col, inv4s2, lags, nlags, out):
        c = ctx_of(h)
        _arr(out, nlags)[:] = c.kde_lag_sums(col, inv4s2, [lags[q] for q in range(nlags)])
        return 0

    def create_aux(h, out):
        aux = HarnessContext(0)
        aux.attach(ctx_of(h))
        out[0] = aux.handle
        return 0

    def destroy_aux(aux):
        _CTX.pop(int(aux), None)
        return 0

    def kopt2d_enqueue(h, B, F, d_hist, neff, do_corr, fallback_t, corr, d_rows, ticket):
        c = ctx_of(h)
        CALLS.append(("kopt2d_enqueue", c.lane, B, F))
        H = np.asarray(buf_of(d_hist).a, dtype=np.float64).reshape(-1, F * F)[:B].copy()
        buf_of(d_rows).a = (H, F, _arr(neff, B).copy(), _arr(do_corr, B, np.int32).copy(), _arr(fallback_t, B).copy(),
                            _arr(corr, B).copy())
        ticket[0] = 0
        return 0

    def kopt2d_finish(h, ha, ticket, B, d_rows, out):
        c = ctx_of(h)
        CALLS.append(("kopt2d_finish", c.lane, B))
        H, F, ne, dc, fb, co = buf_of(d_rows).a
        o = _arr(out, 12 * B).reshape(B, 12)
        for q in range(B):
            o[q] = c.kopt2d_cached(H[q].reshape(F, F), ne[q], dc[q], fb[q], co[q])
        return 0

    def comm_world(h):
        return getattr(ctx_of(h), "comm_world", 0)

    def comm_allreduce_sum(h, inout, count):
        c = ctx_of(h)
        v = _arr(inout, count)
        v[:] = c.comm_allreduce_sum(v.copy())
        return 0

    def stream_priority(h, level):
        CALLS.append(("stream_priority", ctx_of(h).lane, int(level)))
        return 0

    def last_error(h):
        return b"context double: an entry point raised (see the traceback above)"

    impl = dict(locals())
    ops = Ops()
    keep = []
    for name, proto in _OPS:
        fn = proto(_guard(impl[name]) if name != "last_error" else impl[name])
        keep.append(fn)
        setattr(ops, name, fn)
    return ops, keep


_OPS_STRUCT = None


_OPS1D = [
    ("hist1d_dev", C.CFUNCTYPE(C.c_int, _p, _pi32, _i32, _pd, _pd, _i32, _p)),
    ("isj1d_dev", C.CFUNCTYPE(C.c_int, _p, _i32, _i32, _p, _pd, _pd, _pi32)),
    ("density1d_dev", C.CFUNCTYPE(C.c_int, _p, _i32, _i32, _p, _pd, _pi32, _pi32, _i32, _i32, _pd, _pi32)),
    ("fetch", C.CFUNCTYPE(C.c_int, _p, _p, _p, _i64)),
]


class Ops1D(C.Structure):
    _fields_ = _OPS1D


_OPS1D_STRUCT = None


def ops1d_struct():
    """The 1D stages of gd_density1d_batch bound to the numpy context double (device pointers = FakeBuf handles)."""
    global _OPS1D_STRUCT
    if _OPS1D_STRUCT is not None:
        return _OPS1D_STRUCT[0]

    def hist1d_dev(h, cols, ncols, binmin, width, F, d_hist):
        c = _CTX[int(h)]
        CALLS.append(("hist1d_dev", c.lane, ncols))
        _BUF[int(d_hist)].a = c.hist1d([cols[q] for q in range(ncols)], [binmin[q] for q in range(ncols)],
                                       [width[q] for q in range(ncols)], F)
        return 0

    def isj1d_dev(h, B, F, d_hist, neff, hfrac, status):
        c = _CTX[int(h)]
        CALLS.append(("isj1d_dev", c.lane, B))
        hh, st = c.isj1d(np.asarray(_BUF[int(d_hist)].a).reshape(B, F), [neff[b] for b in range(B)])
        for b in range(B):
            hfrac[b], status[b] = hh[b], int(st[b])
        return 0

    def density1d_dev(h, B, F, d_hist, smooth, winw, flags, bco, mbc, P_out, status):
        c = _CTX[int(h)]
        CALLS.append(("density1d_dev", c.lane, B))
        P, st = c.density1d(np.asarray(_BUF[int(d_hist)].a).reshape(B, F), [smooth[b] for b in range(B)],
                            [winw[b] for b in range(B)], [flags[b] for b in range(B)], bco, mbc)
        np.ctypeslib.as_array(P_out, shape=(B * F,))[:] = np.asarray(P).ravel()
        for b in range(B):
            status[b] = int(st[b])
        return 0

    def fetch(h, dst, d_src, nbytes):
        a = np.ascontiguousarray(_BUF[int(d_src)].a, dtype=np.float64).ravel()
        C.memmove(dst, a.ctypes.data, int(nbytes))
        return 0

    fns = dict(hist1d_dev=hist1d_dev, isj1d_dev=isj1d_dev, density1d_dev=density1d_dev, fetch=fetch)
    keep = [t(_guard(fns[name])) for name, t in _OPS1D]
    _OPS1D_STRUCT = (Ops1D(*keep), keep)
    return _OPS1D_STRUCT[0]


def ops_struct():
    global _OPS_STRUCT
    if _OPS_STRUCT is None:
        _OPS_STRUCT = _make_ops()
    return _OPS_STRUCT[0]


_HARNESS = None


def harness():
    global _HARNESS
    if _HARNESS is None:
        import build

        lib = build.load_batch()
        lib.gdt_batch_state_new.restype = _p
        lib.gdt_batch_state_free.argtypes = [_p, _p, _p]
        lib.gdt_batch_finish.argtypes = [_p, _p, _p]
        lib.gdt_batch_invalidate.argtypes = [_p]
        lib.gdt_density2d_batch.argtypes = [_p, _p, _p, _p, _p, _p, _i32, _pd, _pd, _pd, _pi32, _i32, _p, _p, _pd, _i64, _pi32, _pd,
                                            _pd, _pi32, _pi32, C.c_char_p, _i32]
        lib.gdt_grid_sizes.argtypes = [_p, _i32, _pd, _pi32, _i32, _pi32]
        lib.gdt_chol_shear.argtypes = [_f64, _f64, _f64, _pd, _pd]
        lib.gdt_py_pow.argtypes = [_f64, _f64]
        lib.gdt_py_pow.restype = _f64
        _HARNESS = lib
    return _HARNESS


class HarnessContext(FakeContext):
    """The numpy context double plus the batch entry points of getdist_amd._lib.Context, served by batch2d.hpp."""

    _count = [0]

    def __init__(self, device=0):
        super().__init__(device)
        HarnessContext._count[0] += 1
        self.handle = HarnessContext._count[0]
        self.lane = self.handle
        _CTX[self.handle] = self
        self.state = None
        self.h = self.handle  # "not closed" for PendingBatch

    def kopt2d_cached(self, H, neff, do_corr, fallback_t, corr):
        key = (H.tobytes(), float(neff), int(do_corr), float(fallback_t), float(corr))
        if key not in _KOPT_CACHE:
            _KOPT_CACHE[key] = FakeContext.kopt2d(self, FakeBuf(H[None]), 1, H.shape[0], [neff], [do_corr], [fallback_t], [corr])[0]
        return _KOPT_CACHE[key]

    def kopt2d(self, d_hist, B, F, neff, do_corr, fallback_t, corr):  # the Python-planned route shares the cache
        self.log.append(("kopt2d", B, F))
        H = np.asarray(d_hist.a, dtype=np.float64).reshape(B, F, F)
        return np.array([self.kopt2d_cached(H[b], neff[b], do_corr[b], fallback_t[b], corr[b]) for b in range(B)])

    def batch2d_grid_sizes(self, settings, n, corr, pairs32):
        F = np.zeros(len(pairs32), dtype=np.int32)
        harness().gdt_grid_sizes(C.byref(settings), int(n), corr.ctypes.data_as(_pd), pairs32.ctypes.data_as(_pi32), len(pairs32),
                                 F.ctypes.data_as(_pi32))
        return F

    def density2d_batch(self, twin, settings, params, n, corr, cov, lag_probe, pairs32, exchange, grids, status, meta, levels,
                        level_status):
        from getdist_amd._lib import GdhipError

        lib = harness()
        CALLS.append(("density2d_batch", self.lane, len(pairs32), int(settings.results_in_flight)))
        if self.state is None:
            self.state = lib.gdt_batch_state_new()
        tokens = np.full(2, -1, dtype=np.int32)
        err = C.create_string_buffer(512)
        rc = lib.gdt_density2d_batch(
            C.byref(ops_struct()), self.state, self.handle, None if twin is None else twin.handle, C.byref(settings),
            C.cast(params, _p), int(n), corr.ctypes.data_as(_pd), cov.ctypes.data_as(_pd),
            None if lag_probe is None else lag_probe.ctypes.data_as(_pd), pairs32.ctypes.data_as(_pi32), len(pairs32),
            None if exchange is None else C.cast(exchange, _p), None, grids.ctypes.data_as(_pd), int(grids.size),
            status.ctypes.data_as(_pi32), meta.ctypes.data_as(_pd), None if levels is None else levels.ctypes.data_as(_pd),
            None if level_status is None else level_status.ctypes.data_as(_pi32), tokens.ctypes.data_as(_pi32), err, 512)
        if rc != 0:
            raise GdhipError(rc, err.value.decode())
        return int(tokens[0]), int(tokens[1])

    def density1d_batch(self, settings, params, n, cols32, want_hist=False):
        from getdist_amd._lib import GdhipError

        lib = harness()
        if self.state is None:
            self.state = lib.gdt_batch_state_new()
        cols32 = np.ascontiguousarray(cols32, dtype=np.int32)
        B, F = len(cols32), int(settings.fine_bins)
        P = np.zeros((B, F))
        hist = np.zeros((B, F)) if want_hist else None
        meta = np.zeros((B, 9))
        err = C.create_string_buffer(512)
        lib.gdt_density1d_batch.argtypes = [_p, _p, _p, _p, _p, _p, _i32, _pi32, _i32, _pd, _pd, _pd, C.c_char_p, _i32]
        rc = lib.gdt_density1d_batch(C.byref(ops_struct()), C.byref(ops1d_struct()), self.state, self.handle, C.byref(settings),
                                     C.cast(params, _p), int(n), cols32.ctypes.data_as(_pi32), B, P.ctypes.data_as(_pd),
                                     None if hist is None else hist.ctypes.data_as(_pd), meta.ctypes.data_as(_pd), err, 512)
        if rc != 0:
            raise GdhipError(rc, err.value.decode())
        return P, hist, meta

    def batch2d_finish(self):
        if self.state is not None:
            harness().gdt_batch_finish(C.byref(ops_struct()), self.state, self.handle)

    def batch2d_invalidate(self):
        if self.state is not None:
            harness().gdt_batch_invalidate(self.state)

    def batch2d_exchanges(self):
        if self.state is None:
            return 0
        fn = harness().gdt_batch_exchanges
        fn.restype, fn.argtypes = C.c_int64, [C.c_void_p]
        return int(fn(self.state))


class PlainContext(FakeContext):
    """The double WITHOUT the batch entry points (the Python-planned route), sharing the optimiser cache."""

    def __init__(self, device=0):
        super().__init__(device)
        self.lane = 0

    kopt2d_cached = HarnessContext.kopt2d_cached
    kopt2d = HarnessContext.kopt2d
