"""
ctypes binding of libgdhip.so (the C ABI in include/gdhip.h).  This is the thin Python->HIP seam; there
is NO CPU fallback: if the library or a GPU is missing, every compute entry point raises.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgdhip.so")

GD_OK, GD_ERR_BADARG, GD_ERR_NOMEM, GD_ERR_HIP, GD_ERR_EMPTY, GD_ERR_SOLVER, GD_ERR_FFT, GD_ERR_NODEVICE, GD_ERR_TIMEOUT = \
    0, -1, -2, -3, -4, -5, -6, -7, -8


class GdhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgdhip error %d: %s" % (code, msg))
        self.code = code


class GdhipTimeout(GdhipError):
    """GD_ERR_TIMEOUT: a collective or the communicator set-up gave up waiting for a peer; the library has aborted and dropped
    the communicator, so the ranks renegotiate over the host application's own channel (parallel.init_library_comm)."""


def build_native(force=False):
    """Compile libgdhip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        for f in os.listdir(os.path.join(_HERE, "csrc")):
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(_HERE, "csrc", f))
    subprocess.run(["bash", script], check=True)
    return LIB_PATH


_p = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double
_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol include/gdhip.h declares (tests check this)
SIGNATURES = {
    "gd_device_count": (C.c_int, []),
    "gd_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "gd_destroy": (None, [_p]),
    "gd_last_error": (C.c_char_p, [_p]),
    "gd_version": (C.c_char_p, []),
    "gd_device_info": (C.c_int, [_p, _pi64]),
    "gd_sync": (C.c_int, [_p]),
    "gd_dev_alloc": (C.c_int, [_p, _i64, C.POINTER(_p)]),
    "gd_dev_free": (C.c_int, [_p, _p]),
    "gd_memcpy_h2d": (C.c_int, [_p, _p, _p, _i64]),
    "gd_memcpy_d2h": (C.c_int, [_p, _p, _p, _i64]),
    "gd_memcpy_d2h_async": (C.c_int, [_p, _p, _p, _i64]),
    "gd_copy_sync": (C.c_int, [_p]),
    "gd_memcpy_d2d": (C.c_int, [_p, _p, _p, _i64]),
    "gd_memset": (C.c_int, [_p, _p, C.c_int, _i64]),
    "gd_gather_items": (C.c_int, [_p, _p, _p, _pi32, _i32, _i64]),
    "gd_host_alloc": (C.c_int, [_p, _i64, C.POINTER(_p)]),
    "gd_host_free": (C.c_int, [_p, _p]),
    "gd_kde_lag_sums_2d": (C.c_int, [_p, _i32, _i32, _pd, _pi64, _i32, _pd]),
    "gd_autocov_lags_batch": (C.c_int, [_p, _pi32, _i32, _pd, _i64, _i32, _pd]),
    "gd_kde_lag_sums_batch": (C.c_int, [_p, _pi32, _i32, _pd, _pi64, _i32, _pd]),
    "gd_autocov_lags_range_batch": (C.c_int, [_p, _pi32, _i32, _pd, _i64, _i64, _i64, _i32, _pd]),
    "gd_timer_start": (C.c_int, [_p]),
    "gd_timer_stop_ms": (C.c_int, [_p, _pd]),
    "gd_upload": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _p]),
    "gd_num_rows": (C.c_int, [_p, _pi64, _pi64]),
    "gd_column_ptr": (C.c_int, [_p, _i64, C.POINTER(_p)]),
    "gd_weight_stats": (C.c_int, [_p, _i64, _i64, _f64, _pd]),
    "gd_col_stats": (C.c_int, [_p, _i64, _i64, _pd]),
    "gd_cov": (C.c_int, [_p, _pi32, _i32, _i64, _i64, _pd, _pd, _pd, _pd]),
    "gd_quantiles": (C.c_int, [_p, _pi32, _i32, _i64, _i64, _pd, _i32, _pd]),
    "gd_quantiles_mm_probe": (C.c_int, [_p, _pi32, _i32, _i64, _i64, _pd, _i32, _pd, _pd, _pd, _pd, C.POINTER(C.c_int32)]),
    "gd_quantiles_mm": (C.c_int, [_p, _pi32, _i32, _i64, _i64, _pd, _i32, _pd, _pd]),
    "gd_autocov_lags": (C.c_int, [_p, _i32, _f64, _i64, _i32, _pd]),
    "gd_kde_lag_sums": (C.c_int, [_p, _i32, _f64, _pi64, _i32, _pd]),
    "gd_hist1d": (C.c_int, [_p, _pi32, _i32, _pd, _pd, _i32, _pd]),
    "gd_hist1d_dev": (C.c_int, [_p, _pi32, _i32, _pd, _pd, _i32, _p]),
    "gd_bin_indices": (C.c_int, [_p, _i32, _f64, _f64, _i32, _i32, _pi32, _pi64]),
    "gd_prebin": (C.c_int, [_p, _i32, _f64, _f64, _i32, _p]),
    "gd_prebin_batch": (C.c_int, [_p, _pi32, _i32, _pd, _pd, _i32, C.POINTER(_p)]),
    "gd_prebin8_batch": (C.c_int, [_p, _pi32, _i32, _pd, _pd, _i32, C.POINTER(_p), _pi64]),
    "gd_hist2d_prebinned8": (C.c_int, [_p, _i32, C.POINTER(_p), C.POINTER(_p), _p]),
    "gd_prebin8_hist2d": (C.c_int, [_p, _pi32, _i32, _pd, _pd, C.POINTER(_p), _pi64, _i32, C.POINTER(_p), C.POINTER(_p), _p]),
    "gd_hist2d": (C.c_int, [_p, _i32, _pi32, _pi32, _pd, _pd, _pd, _pd, _i32, _p]),
    "gd_hist2d_prebinned": (C.c_int, [_p, _i32, C.POINTER(_p), C.POINTER(_p), _i32, _p]),
    "gd_minmax_affine": (C.c_int, [_p, _i32, _pi32, _pi32, _pd, _pd, _pd]),
    "gd_hist2d_sheared": (C.c_int, [_p, _i32, _pi32, _pi32, _pd, _pd, _pd, _pd, _pd, _pd, _i32, _p]),
    "gd_dct1d": (C.c_int, [_p, _i32, _i32, _pd, _pd]),
    "gd_isj1d": (C.c_int, [_p, _i32, _i32, _pd, _pd, _pd, _pi32]),
    "gd_isj1d_dev": (C.c_int, [_p, _i32, _i32, _p, _pd, _pd, _pi32]),
    "gd_density1d": (C.c_int, [_p, _i32, _i32, _pd, _pd, _pi32, _pi32, _i32, _i32, _pd, _pi32]),
    "gd_density1d_dev": (C.c_int, [_p, _i32, _i32, _p, _pd, _pi32, _pi32, _i32, _i32, _pd, _pi32]),
    "gd_density1d_batch": (C.c_int, [_p, _p, _p, _i32, _pi32, _i32, _pd, _pd, _pd]),
    "gd_kopt2d": (C.c_int, [_p, _i32, _i32, _p, _pd, _pi32, _pd, _pd, _pd]),
    "gd_kopt2d_enqueue": (C.c_int, [_p, _i32, _i32, _p, _pd, _pi32, _pd, _pd, _p, _pi32]),
    "gd_kopt2d_finish": (C.c_int, [_p, _p, _i32, _i32, _p, _pd]),
    "gd_get_h": (C.c_int, [_p, _i32, _pd, _pd, _pd, _pi32, _pd]),
    "gd_density2d": (C.c_int, [_p, _i32, _i32, _p, _pd, _pd, _pd, _pi32, _pi32, _i32, _i32, _p, _pi32]),
    "gd_copy_mark": (C.c_int, [_p, _pi32]),
    "gd_copy_wait": (C.c_int, [_p, _i32]),
    "gd_density2d_enqueue": (C.c_int, [_p, _i32, _i32, _p, _pd, _pd, _pd, _pi32, _pi32, _i32, _i32, _p, _p]),
    "gd_density2d_enqueue_indexed": (C.c_int, [_p, _i32, _i32, _p, _pi32, _pd, _pd, _pd, _pi32, _pi32, _i32, _i32, _p, _p]),
    "gd_attach_samples": (C.c_int, [_p, _p]),
    "gd_bind_thread": (C.c_int, [_p]),
    "gd_contour_levels": (C.c_int, [_p, _i32, _i32, _p, _pd, _i32, _pd, _pi32]),
    "gd_limits1d": (C.c_int, [_p, _i32, _i32, _pd, _pd, _pd, _pd, _i32, _i32, _pd, _pi32]),
    "gd_set_extra_column": (C.c_int, [_p, _i32, _pd]),
    "gd_aux_weights": (C.c_int, [_p, _pd]),
    "gd_col_minmax": (C.c_int, [_p, _pi32, _i32, _i64, _i64, _i32, C.c_double, _pd]),
    "gd_weights_integral": (C.c_int, [_p, _pi32]),
    "gd_thin_rows": (C.c_int, [_p, _i64, _i64, _i64, _i32, _p, _i64, C.POINTER(C.c_int64)]),
    "gd_binary_transitions": (C.c_int, [_p, _pi32, _i32, _p, _i64, _pd, _i32, C.POINTER(C.c_int64)]),
    "gd_thinned_lag_sums": (C.c_int, [_p, _pi32, _i32, _pd, _p, _i64, _i32, _pd]),
    "gd_density2d_masked": (C.c_int, [_p, _i32, _p, C.c_double, C.c_double, C.c_double, _i32, _i32, _i32, _i32, _pd, _pd,
                                      C.POINTER(C.c_ubyte), _p, _pi32]),
    "gd_like_weights": (C.c_int, [_p, _pd, _i32, C.c_double, _pd]),
    "gd_select_weights": (C.c_int, [_p, _i32]),
    "gd_likes1d": (C.c_int, [_p, _i32, _i32, _pd, _pd, _pd, _pd, _pi32, _pi32, _i32, _pd, _pi32]),
    "gd_likes2d": (C.c_int, [_p, _i32, _i32, _p, _p, _pd, _pd, _pd, _pi32, _pi32, _i32, _p, _pi32]),
    "gd_circ_convolve": (C.c_int, [_p, _i32, _i32, _pd, _pd, _pd]),
    "gd_convolve1d_direct": (C.c_int, [_p, _pd, _i64, _pd, _i64, _pd]),
    "gd_autoconvolve": (C.c_int, [_p, _i32, _f64, _i32, _pd, _i64, _i64, _i64, _i32, _pd]),
    "gd_like_stats": (C.c_int, [_p, _i32, _pd]),
    "gd_batch2d_grid_sizes": (C.c_int, [_p, _i32, _pd, _pi32, _i32, _pi32]),
    "gd_density2d_batch": (C.c_int, [_p, _p, _p, _p, _i32, _pd, _pd, _pd, _pi32, _i32, _p, _p, _p, _i64, _pi32, _pd, _pd, _pi32,
                                     _pi32]),
    "gd_comm_unique_id": (C.c_int, [_p]),
    "gd_comm_init": (C.c_int, [_p, _i32, _i32, _p]),
    "gd_comm_info": (C.c_int, [_p, _pi32, _pi32]),
    "gd_comm_destroy": (C.c_int, [_p]),
    "gd_comm_abandon": (C.c_int, [_p]),
    "gd_comm_allgather": (C.c_int, [_p, _pd, _i64, _pd]),
    "gd_comm_allreduce_sum": (C.c_int, [_p, _pd, _i64]),
    "gd_comm_allgather_dev": (C.c_int, [_p, _p, _i64, _p]),
    "gd_comm_allreduce_sum_dev": (C.c_int, [_p, _p, _i64, _p]),
    "gd_batch2d_finish": (C.c_int, [_p]),
    "gd_batch2d_invalidate": (C.c_int, [_p]),
    "gd_batch2d_exchanges": (C.c_int, [_p, C.POINTER(C.c_int64)]),
    "gd_comm_rccl_path": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "gd_upload_shard": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "gd_comm_share_columns": (C.c_int, [_p, C.POINTER(C.c_int64)]),
}

_lib = None


def load_library():
    """dlopen libgdhip.so and attach prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libgdhip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or getdist_amd/csrc/build.sh (there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _dp(a):
    return a.ctypes.data_as(_pd)


def _ip(a):
    return a.ctypes.data_as(_pi32)


def _f64arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32arr(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class DevBuf:
    """
    A device allocation owned by a Context.  free() returns the block to the context's free list (hipFree
    synchronises the device and costs ~0.25 ms; a triangle makes ~100 short-lived buffers per step).
    """

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr, self.capacity = ctx._take_block(self.nbytes)

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx._give_block(self.ptr, self.capacity)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def to_host(self, shape, dtype=np.float64, offset_bytes=0, pinned=False):
        out = self.ctx.pinned_array(shape, dtype) if pinned else np.empty(shape, dtype=dtype)
        self.ctx._check(self.ctx.lib.gd_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr + offset_bytes, out.nbytes))
        return out

    def to_host_async(self, shape, dtype=np.float64):
        """Start a D2H copy into page-locked memory on the copy stream; call ctx.copy_sync() before reading."""
        out = self.ctx.pinned_array(shape, dtype)
        self.ctx._check(self.ctx.lib.gd_memcpy_d2h_async(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def from_host(self, arr, offset_bytes=0):
        arr = np.ascontiguousarray(arr)
        self.ctx._check(self.ctx.lib.gd_memcpy_h2d(self.ctx.h, self.ptr + offset_bytes, arr.ctypes.data, arr.nbytes))


class Context:
    """One GPU, one stream, one resident sample set.  Thin, typed wrappers over the C ABI."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = None
        if self.lib.gd_device_count() <= 0:
            raise RuntimeError("getdist_amd needs a HIP device (no CPU fallback): none visible")
        h = _p()
        rc = self.lib.gd_create(int(device), C.byref(h))
        if rc != 0:
            raise GdhipError(rc, "gd_create(device=%d) failed" % device)
        self.h = h
        self.device = device
        self.N = self.n = 0
        self.weighted = False
        self._pinned = []  # (ptr, nbytes, ctypes buffer): page-locked result buffers, recycled when unreferenced
        self._free_blocks = []  # (device ptr, capacity) returned by DevBuf.free(), reused by alloc()
        self._pinned_blocks = []
        self.comm_world = self.comm_rank = 0

    PINNED_POOL_LIMIT = 8 << 30

    def pinned_array(self, shape, dtype=np.float64):
        """
        A numpy array in page-locked host memory (D2H at full PCIe rate).  Blocks are recycled once no array
        view references them any more (every view keeps the block's ctypes buffer alive, so its refcount tells).
        """
        import sys

        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        best = None
        for k, (ptr, nb, buf) in enumerate(self._pinned):
            if nb >= nbytes and sys.getrefcount(buf) <= 3 and (best is None or nb < self._pinned[best][1]):
                best = k
        if best is None:
            if sum(nb for _, nb, _ in self._pinned) + nbytes > self.PINNED_POOL_LIMIT:
                return np.empty(shape, dtype=dtype)
            p = _p()
            self._check(self.lib.gd_host_alloc(self.h, max(nbytes, 1 << 20), C.byref(p)))
            nb = max(nbytes, 1 << 20)
            buf = (C.c_ubyte * nb).from_address(p.value)
            self._pinned.append((p.value, nb, buf))
            best = len(self._pinned) - 1
        buf = self._pinned[best][2]
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def pinned_block(self, shape, dtype=np.float64):
        """A page-locked array outside the recycling pool (freed with the context): the landing buffer of the binary
        chain cache, from which the sample columns are DMA'd to the device."""
        nbytes = max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 8)
        p = _p()
        self._check(self.lib.gd_host_alloc(self.h, nbytes, C.byref(p)))
        buf = (C.c_ubyte * nbytes).from_address(p.value)
        self._pinned_blocks.append((p.value, buf))
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def reserve_pinned_twin(self):
        """Allocate a second block for every pooled page-locked block (callers that keep one result set alive while
        computing the next need two sets; doing it up front keeps hipHostMalloc out of the steady state)."""
        for ptr, nb, buf in list(self._pinned):
            p = _p()
            self._check(self.lib.gd_host_alloc(self.h, nb, C.byref(p)))
            self._pinned.append((p.value, nb, (C.c_ubyte * nb).from_address(p.value)))

    def close(self):
        """Free the device side.  Page-locked host blocks that some numpy array still views (mc.samples loaded from the
        binary cache, a Density2D.P the user kept) are NOT freed: every view holds the block's ctypes buffer, so its
        reference count tells; such a block stays valid for the life of the process instead of dangling."""
        import sys

        if self.h:
            self.lib.gd_copy_sync(self.h)  # result copies in flight land before their targets could go away
            self.release_cached_blocks()
            for ptr, _, buf in getattr(self, "_pinned", []):
                if sys.getrefcount(buf) <= 3:
                    self.lib.gd_host_free(self.h, ptr)
            self._pinned = []
            for ptr, buf in getattr(self, "_pinned_blocks", []):
                if sys.getrefcount(buf) <= 3:
                    self.lib.gd_host_free(self.h, ptr)
            self._pinned_blocks = []
            self.lib.gd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.gd_last_error(self.h)
            if rc == GD_ERR_TIMEOUT:
                self.comm_world = self.comm_rank = 0  # the library has aborted and dropped the communicator
                raise GdhipTimeout(rc, msg.decode() if msg else "?")
            raise GdhipError(rc, msg.decode() if msg else "?")

    # ---- memory / info
    def device_info(self):
        a = np.zeros(6, dtype=np.int64)
        self._check(self.lib.gd_device_info(self.h, a.ctypes.data_as(_pi64)))
        return dict(cu_count=int(a[0]), lds_bytes=int(a[1]), hbm_total=int(a[2]), hbm_free=int(a[3]),
                    clock_khz=int(a[4]), wave=int(a[5]))

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    DEVICE_CACHE_LIMIT = 16 << 30

    def _take_block(self, nbytes):
        best = None
        for k, (ptr, cap) in enumerate(self._free_blocks):
            if cap >= nbytes and cap <= 2 * nbytes + (1 << 20) and (best is None or cap < self._free_blocks[best][1]):
                best = k
        if best is not None:
            return self._free_blocks.pop(best)
        p = _p()
        rc = self.lib.gd_dev_alloc(self.h, int(nbytes), C.byref(p))
        if rc == GD_ERR_NOMEM and self._free_blocks:  # give cached blocks back and retry once
            self.release_cached_blocks()
            rc = self.lib.gd_dev_alloc(self.h, int(nbytes), C.byref(p))
        self._check(rc)
        return p.value, int(nbytes)

    def _give_block(self, ptr, cap):
        if sum(c for _, c in self._free_blocks) + cap > self.DEVICE_CACHE_LIMIT:
            self.lib.gd_dev_free(self.h, ptr)
        else:
            self._free_blocks.append((ptr, cap))

    def release_cached_blocks(self):
        for ptr, _ in self._free_blocks:
            self.lib.gd_dev_free(self.h, ptr)
        self._free_blocks = []

    def sync(self):
        self._check(self.lib.gd_sync(self.h))

    def gather_items(self, dst, src, index, item_bytes, dst_offset=0):
        """dst[dst_offset + q] = src[index[q]] for fixed-size items (one gather kernel)."""
        index = _i32arr(index)
        self._check(self.lib.gd_gather_items(self.h, dst.ptr + int(dst_offset) * int(item_bytes), src.ptr, _ip(index),
                                             len(index), int(item_bytes)))

    def autocov_lags_batch(self, cols, means, k0, nlags):
        cols, means = _i32arr(cols), _f64arr(means)
        out = np.zeros((len(cols), nlags))
        self._check(self.lib.gd_autocov_lags_batch(self.h, _ip(cols), len(cols), _dp(means), int(k0), int(nlags), _dp(out)))
        return out

    def autocov_lags_range_batch(self, cols, means, lo, hi, k0, nlags):
        cols, means = _i32arr(cols), _f64arr(means)
        out = np.zeros((len(cols), nlags))
        self._check(self.lib.gd_autocov_lags_range_batch(self.h, _ip(cols), len(cols), _dp(means), int(lo), int(hi), int(k0),
                                                         int(nlags), _dp(out)))
        return out

    def kde_lag_sums_2d(self, coli, colj, kinv3, lags):
        kinv3 = _f64arr(kinv3)
        lags = np.ascontiguousarray(lags, dtype=np.int64)
        out = np.zeros(len(lags))
        self._check(self.lib.gd_kde_lag_sums_2d(self.h, int(coli), int(colj), _dp(kinv3), lags.ctypes.data_as(_pi64),
                                                len(lags), _dp(out)))
        return out

    def kde_lag_sums_batch(self, cols, inv4s2, lags):
        cols, inv4s2 = _i32arr(cols), _f64arr(inv4s2)
        lags = np.ascontiguousarray(lags, dtype=np.int64)
        out = np.zeros((len(cols), len(lags)))
        self._check(self.lib.gd_kde_lag_sums_batch(self.h, _ip(cols), len(cols), _dp(inv4s2), lags.ctypes.data_as(_pi64),
                                                   len(lags), _dp(out)))
        return out

    def copy_sync(self):
        self._check(self.lib.gd_copy_sync(self.h))

    def copy_mark(self):
        """A token for 'every result copy issued so far'; copy_wait(token) blocks until those have landed."""
        tok = C.c_int32(0)
        self._check(self.lib.gd_copy_mark(self.h, C.byref(tok)))
        return int(tok.value)

    def copy_wait(self, token):
        self._check(self.lib.gd_copy_wait(self.h, int(token)))

    def copy_d2d(self, dst, dst_off, src, src_off, nbytes):
        self._check(self.lib.gd_memcpy_d2d(self.h, dst.ptr + dst_off, src.ptr + src_off, int(nbytes)))

    def timer_start(self):
        self._check(self.lib.gd_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_double()
        self._check(self.lib.gd_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- sample set
    def upload(self, samples, weights=None):
        s = np.asarray(samples)
        if s.dtype != np.float64:
            s = s.astype(np.float64)
        if s.ndim == 1:
            s = s.reshape(-1, 1)
        if not (s.flags.c_contiguous or s.flags.f_contiguous):
            s = np.ascontiguousarray(s)
        N, n = s.shape
        rs, cs = s.strides[0] // 8, s.strides[1] // 8
        if n == 1:
            rs, cs = 1, N
        w = None if weights is None else _f64arr(weights)
        self._keep = (s, w)
        self._check(self.lib.gd_upload(self.h, s.ctypes.data, N, n, rs, cs, None if w is None else w.ctypes.data))
        self._keep = None
        self.N, self.n, self.weighted = N, n, w is not None

    def upload_shard(self, cols_f, N, n, col_first, weights=None):
        """This rank's block of columns only (``cols_f``: (N, count) Fortran-ordered fp64, the columns
        [col_first, col_first + count) of the (N, n) set); the rest arrives by comm_share_columns."""
        c = np.asfortranarray(cols_f, dtype=np.float64)
        if c.ndim == 1:
            c = c.reshape(-1, 1, order="F")
        count = c.shape[1]
        assert c.shape[0] == N or count == 0
        w = None if weights is None else _f64arr(weights)
        self._keep = (c, w)
        self._check(self.lib.gd_upload_shard(self.h, c.ctypes.data if count else None, int(N), int(n), int(col_first), int(count),
                                             int(c.strides[1] // 8) if count else int(N), None if w is None else w.ctypes.data))
        self._keep = None
        self.N, self.n, self.weighted = int(N), int(n), w is not None

    def comm_share_columns(self, first_by_rank):
        f = np.ascontiguousarray(first_by_rank, dtype=np.int64)
        assert f.size == self.comm_world + 1
        self._check(self.lib.gd_comm_share_columns(self.h, f.ctypes.data_as(C.POINTER(C.c_int64))))

    @staticmethod
    def comm_rccl_path():
        lib = load_library()
        buf = C.create_string_buffer(1024)
        pre = C.c_int32(0)
        rc = lib.gd_comm_rccl_path(buf, 1024, C.byref(pre))
        if rc != 0:
            raise GdhipError(rc, "librccl.so could not be loaded")
        return buf.value.decode(), bool(pre.value)

    def attach(self, owner):
        """Borrow ``owner``'s resident sample set (same device, no copy): this context becomes a second lane."""
        self._check(self.lib.gd_attach_samples(self.h, owner.h))
        self.N, self.n, self.weighted = owner.N, owner.n, owner.weighted

    def bind_thread(self):
        self._check(self.lib.gd_bind_thread(self.h))

    def column_ptr(self, j):
        p = _p()
        self._check(self.lib.gd_column_ptr(self.h, int(j), C.byref(p)))
        return p.value

    # ---- moments
    def weight_stats(self, lo=0, hi=None, thresh=np.inf):
        out = np.zeros(4)
        self._check(self.lib.gd_weight_stats(self.h, lo, self.N if hi is None else hi, float(thresh), _dp(out)))
        return dict(norm=out[0], max_w=out[1], sum_w2=out[2], n_above=out[3])

    def col_stats(self, lo=0, hi=None):
        out = np.zeros((self.n, 4))
        self._check(self.lib.gd_col_stats(self.h, lo, self.N if hi is None else hi, _dp(out)))
        return out

    def cov(self, cols=None, lo=0, hi=None, minmax=False):
        """(means, cov, norm) of the columns over rows [lo, hi); with ``minmax`` also the (m, 2) column extrema."""
        cols = _i32arr(np.arange(self.n) if cols is None else cols)
        m = len(cols)
        means, cov, norm = np.zeros(m), np.zeros((m, m)), C.c_double()
        mm = np.zeros((m, 2)) if minmax else None
        self._check(self.lib.gd_cov(self.h, _ip(cols), m, lo, self.N if hi is None else hi, _dp(means), _dp(cov),
                                    C.byref(norm), None if mm is None else _dp(mm)))
        return (means, cov, norm.value, mm) if minmax else (means, cov, norm.value)

    def quantiles(self, cols, targets, lo=0, hi=None, minmax=None):
        """Weighted quantile selection; ``minmax`` (len(cols) x 2: each column's min and max over the rows, or over a
        superset of them) enables the two-pass linear-bucket path."""
        cols = _i32arr(cols)
        targets = _f64arr(targets).reshape(len(cols), -1)
        out = np.zeros_like(targets)
        mm = None if minmax is None else _f64arr(minmax).reshape(len(cols), 2)
        self._check(self.lib.gd_quantiles_mm(self.h, _ip(cols), len(cols), lo, self.N if hi is None else hi, _dp(targets),
                                             targets.shape[1], None if mm is None else _dp(mm), _dp(out)))
        return out

    def quantiles_probe(self, cols, targets, minmax, means):
        """gd_quantiles_mm_probe over whole columns: (quantiles, lag probe or None).  The probe -- the first 8
        autocovariance lag sums of the columns about ``means``, what autocov_lags_batch(cols, means, 0, 8) returns -- rides on
        the select's counting pass; None when the select took a path without it."""
        cols = _i32arr(cols)
        targets = _f64arr(targets).reshape(len(cols), -1)
        out = np.zeros_like(targets)
        mm = _f64arr(minmax).reshape(len(cols), 2)
        means = _f64arr(means)
        probe = np.zeros((len(cols), 8))
        done = C.c_int32(0)
        self._check(self.lib.gd_quantiles_mm_probe(self.h, _ip(cols), len(cols), 0, self.N, _dp(targets), targets.shape[1], _dp(mm),
                                                   _dp(out), _dp(means), _dp(probe), C.byref(done)))
        return out, (probe if done.value else None)

    def autocov_lags(self, col, mean, k0, nlags):
        out = np.zeros(nlags)
        self._check(self.lib.gd_autocov_lags(self.h, int(col), float(mean), int(k0), int(nlags), _dp(out)))
        return out

    def kde_lag_sums(self, col, inv4s2, lags):
        lags = np.ascontiguousarray(lags, dtype=np.int64)
        out = np.zeros(len(lags))
        self._check(self.lib.gd_kde_lag_sums(self.h, int(col), float(inv4s2), lags.ctypes.data_as(_pi64), len(lags),
                                             _dp(out)))
        return out

    # ---- binning
    def hist1d(self, cols, binmin, width, F):
        cols, binmin, width = _i32arr(cols), _f64arr(binmin), _f64arr(width)
        out = np.zeros((len(cols), F))
        self._check(self.lib.gd_hist1d(self.h, _ip(cols), len(cols), _dp(binmin), _dp(width), int(F), _dp(out)))
        return out

    def bin_indices(self, col, binmin, width, F, round_half=True):
        idx = np.zeros(self.N, dtype=np.int32)
        bad = C.c_int64()
        self._check(self.lib.gd_bin_indices(self.h, int(col), float(binmin), float(width), int(bool(round_half)), int(F),
                                            _ip(idx), C.byref(bad)))
        return idx, bad.value

    def prebin(self, col, binmin, width, F, buf=None):
        buf = buf or self.alloc(self.N * 2 + 64)
        self._check(self.lib.gd_prebin(self.h, int(col), float(binmin), float(width), int(F), buf.ptr))
        return buf

    def prebin_batch(self, cols, binmin, width, F, bufs):
        cols, binmin, width = _i32arr(cols), _f64arr(binmin), _f64arr(width)
        arr = (_p * len(cols))(*[b.ptr for b in bufs])
        self._check(self.lib.gd_prebin_batch(self.h, _ip(cols), len(cols), _dp(binmin), _dp(width), int(F), arr))

    def prebin8_batch(self, cols, binmin, width, F, bufs):
        """Byte index columns (F <= 256) for several sample columns in one launch; returns the out-of-range counts."""
        cols, binmin, width = _i32arr(cols), _f64arr(binmin), _f64arr(width)
        arr = (_p * len(cols))(*[b.ptr for b in bufs])
        bad = np.zeros(len(cols), dtype=np.int64)
        self._check(self.lib.gd_prebin8_batch(self.h, _ip(cols), len(cols), _dp(binmin), _dp(width), int(F), arr,
                                              bad.ctypes.data_as(_pi64)))
        return bad

    def hist2d_prebinned8(self, idx_x, idx_y, out=None):
        """B histograms of 256 x 256 bins from byte index columns (unit weights); raises GdhipError(-5) on counter wrap."""
        B = len(idx_x)
        out = out or self.alloc(B * 65536 * 8)
        if isinstance(idx_x, np.ndarray):  # device addresses already gathered (uint64), e.g. from a per-column table
            px, py = np.ascontiguousarray(idx_x, dtype=np.uint64), np.ascontiguousarray(idx_y, dtype=np.uint64)
            ax, ay = px.ctypes.data_as(C.POINTER(_p)), py.ctypes.data_as(C.POINTER(_p))
        else:
            ax = (_p * B)(*[b.ptr for b in idx_x])
            ay = (_p * B)(*[b.ptr for b in idx_y])
        self._check(self.lib.gd_hist2d_prebinned8(self.h, B, ax, ay, out.ptr))
        return out

    def hist2d(self, colx, coly, bx, wx, by, wy, F, out=None):
        colx, coly = _i32arr(colx), _i32arr(coly)
        B = len(colx)
        out = out or self.alloc(B * F * F * 8)
        bx, wx, by, wy = _f64arr(bx), _f64arr(wx), _f64arr(by), _f64arr(wy)
        self._check(self.lib.gd_hist2d(self.h, B, _ip(colx), _ip(coly), _dp(bx), _dp(wx), _dp(by), _dp(wy), int(F),
                                       out.ptr))
        return out

    def hist2d_prebinned(self, idx_x, idx_y, F, out=None):
        B = len(idx_x)
        out = out or self.alloc(B * F * F * 8)
        ax = (_p * B)(*[b.ptr for b in idx_x])
        ay = (_p * B)(*[b.ptr for b in idx_y])
        self._check(self.lib.gd_hist2d_prebinned(self.h, B, ax, ay, int(F), out.ptr))
        return out

    def minmax_affine(self, coli, colj, a, b):
        coli, colj, a, b = _i32arr(coli), _i32arr(colj), _f64arr(a), _f64arr(b)
        out = np.zeros((len(coli), 2))
        self._check(self.lib.gd_minmax_affine(self.h, len(coli), _ip(coli), _ip(colj), _dp(a), _dp(b), _dp(out)))
        return out

    def hist2d_sheared(self, coli, colj, r0, r1, xmin, dx, ymin, dy, F, out=None):
        coli, colj = _i32arr(coli), _i32arr(colj)
        B = len(coli)
        out = out or self.alloc(B * F * F * 8)
        arrs = [_f64arr(v) for v in (r0, r1, xmin, dx, ymin, dy)]
        self._check(self.lib.gd_hist2d_sheared(self.h, B, _ip(coli), _ip(colj), *[_dp(v) for v in arrs], int(F),
                                               out.ptr))
        return out

    # ---- densities
    def dct1d(self, hist):
        hist = _f64arr(hist)
        B, F = hist.shape
        out = np.zeros_like(hist)
        self._check(self.lib.gd_dct1d(self.h, B, F, _dp(hist), _dp(out)))
        return out

    def isj1d(self, hist, neff):
        """(hfrac[B], status[B]): the 1D ISJ bandwidths of B histograms, solved on the device."""
        hist, neff = _f64arr(hist), _f64arr(neff)
        B, F = hist.shape
        h = np.zeros(B)
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_isj1d(self.h, B, F, _dp(hist), _dp(neff), _dp(h), _ip(status)))
        return h, status

    def density1d(self, hist, smooth, winw, flags, bco, mbc):
        hist = _f64arr(hist)
        B, F = hist.shape
        smooth, winw, flags = _f64arr(smooth), _i32arr(winw), _i32arr(flags)
        P = np.zeros_like(hist)
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_density1d(self.h, B, F, _dp(hist), _dp(smooth), _ip(winw), _ip(flags), int(bco),
                                          int(mbc), _dp(P), _ip(status)))
        return P, status

    def density2d_masked(self, d_hist, pos, F, rx, ry, corr, winw, flags, bco, mbc, mask_bc, mask_mbc, zero_mask):
        """Pair ``pos`` of the histogram batch ``d_hist`` with an explicit prior mask (mask_function);
        returns (device grid, status)."""
        out = self.alloc(F * F * 8)
        status = np.zeros(1, dtype=np.int32)
        mb = None if mask_bc is None else _f64arr(mask_bc)
        mm = None if mask_mbc is None else _f64arr(mask_mbc)
        zm = None if zero_mask is None else np.ascontiguousarray(zero_mask, dtype=np.uint8)
        self._check(self.lib.gd_density2d_masked(
            self.h, int(F), d_hist.ptr + int(pos) * F * F * 8, float(rx), float(ry), float(corr),
            int(winw), int(flags), int(bco), int(mbc), None if mb is None else _dp(mb), None if mm is None else _dp(mm),
            None if zm is None else zm.ctypes.data_as(C.POINTER(C.c_ubyte)), out.ptr, _ip(status)))
        return out, status

    # ---- auxiliary vectors
    EXTRA_COLS = 4

    def set_extra_column(self, slot, x):
        """Copy a host vector into spare column ``slot``; returns the column index (n + slot) it is addressed by."""
        x = _f64arr(x)
        if x.shape != (self.N,):
            raise ValueError("vector must have one entry per sample row")
        self._check(self.lib.gd_set_extra_column(self.h, int(slot), _dp(x)))
        return self.n + int(slot)

    def aux_weights(self, w):
        w = _f64arr(w)
        if w.shape != (self.N,):
            raise ValueError("weights must have one entry per sample row")
        self._check(self.lib.gd_aux_weights(self.h, _dp(w)))

    def col_minmax(self, cols, lo=0, hi=None, cond_col=-1, cond_below=0.0):
        """min / max per column over the rows whose ``cond_col`` value is < ``cond_below`` (cond_col < 0: all rows)."""
        cols = _i32arr(cols)
        out = np.zeros((len(cols), 2))
        self._check(self.lib.gd_col_minmax(self.h, _ip(cols), len(cols), lo, self.N if hi is None else hi,
                                           int(cond_col), float(cond_below), _dp(out)))
        return out

    # ---- contour levels
    def contour_levels(self, d_P, B, F, contours):
        contours = _f64arr(contours)
        out = np.zeros((B, len(contours)))
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_contour_levels(self.h, int(B), int(F), d_P.ptr, _dp(contours), len(contours), _dp(out),
                                               _ip(status)))
        return out, status

    def density2d_enqueue(self, d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc, status):
        """density2d without the final wait: ``status`` is a page-locked int32 array of length B (pinned_array) that is
        valid after sync() / copy_sync() of a later to_host_async; the returned grids likewise."""
        out = self.alloc(B * F * F * 8)
        rx, ry, corr, winw, flags = _f64arr(rx), _f64arr(ry), _f64arr(corr), _i32arr(winw), _i32arr(flags)
        assert status.dtype == np.int32 and status.size == B and status.flags.c_contiguous
        self._check(self.lib.gd_density2d_enqueue(self.h, B, F, d_hist.ptr if isinstance(d_hist, DevBuf) else d_hist,
                                                  _dp(rx), _dp(ry), _dp(corr), _ip(winw), _ip(flags), int(bco), int(mbc),
                                                  out.ptr, status.ctypes.data))
        return out

    # ---- credible limits
    def limits1d(self, P, x0, spacing, contours, factor=0):
        """(limits[B, nc, 4] = lower, upper, has_min, has_top; status[B]) of B densities P[B, F] on regular grids."""
        P, x0, spacing, contours = _f64arr(P), _f64arr(x0), _f64arr(spacing), _f64arr(contours)
        B, F = P.shape
        out = np.zeros((B, len(contours), 4))
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_limits1d(self.h, B, F, _dp(P), _dp(x0), _dp(spacing), _dp(contours), len(contours),
                                         int(factor), _dp(out), _ip(status)))
        return out, status

    # ---- thinned chains
    def weights_integral(self):
        out = C.c_int32()
        self._check(self.lib.gd_weights_integral(self.h, C.byref(out)))
        return bool(out.value)

    def thin_rows(self, lo, hi, factor, unique_mode, capacity):
        """Thinned row list (device int32 buffer, count) of chain rows [lo,hi); see gd_thin_rows."""
        buf = self.alloc(max(int(capacity), 1) * 4)
        n = C.c_int64()
        self._check(self.lib.gd_thin_rows(self.h, int(lo), int(hi), int(factor), int(bool(unique_mode)), buf.ptr,
                                          int(capacity), C.byref(n)))
        return buf, n.value

    def binary_transitions(self, cols, rows, K, thresholds):
        cols = _i32arr(cols)
        thresholds = _f64arr(thresholds).reshape(len(cols), -1)
        out = np.zeros((len(cols), thresholds.shape[1], 12), dtype=np.int64)
        self._check(self.lib.gd_binary_transitions(self.h, _ip(cols), len(cols), rows.ptr, int(K), _dp(thresholds),
                                                   thresholds.shape[1], out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def thinned_lag_sums(self, cols, means, rows, K, maxoff):
        cols, means = _i32arr(cols), _f64arr(means)
        out = np.zeros((len(cols), int(maxoff)))
        self._check(self.lib.gd_thinned_lag_sums(self.h, _ip(cols), len(cols), _dp(means), rows.ptr, int(K), int(maxoff),
                                                 _dp(out)))
        return out

    # ---- mean likelihoods
    def like_weights(self, loglikes, mode, mean_loglike):
        """Build (or with loglikes=None drop) the device vector w*exp(mean_loglike-loglikes) / w*loglikes."""
        if loglikes is None:
            self._check(self.lib.gd_like_weights(self.h, None, 0, 0.0, None))
            return None
        ll = _f64arr(loglikes)
        if ll.shape != (self.N,):
            raise ValueError("loglikes must have one entry per sample row")
        tot = C.c_double()
        self._check(self.lib.gd_like_weights(self.h, _dp(ll), int(mode), float(mean_loglike), C.byref(tot)))
        return tot.value

    def select_weights(self, which):
        self._check(self.lib.gd_select_weights(self.h, int(which)))

    def likes1d(self, hist, likehist, P, smooth, winw, flags, shade_mean_loglikes):
        hist, likehist, P = _f64arr(hist), _f64arr(likehist), _f64arr(P)
        B, F = hist.shape
        smooth, winw, flags = _f64arr(smooth), _i32arr(winw), _i32arr(flags)
        out = np.zeros_like(hist)
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_likes1d(self.h, B, F, _dp(hist), _dp(likehist), _dp(P), _dp(smooth), _ip(winw),
                                        _ip(flags), int(bool(shade_mean_loglikes)), _dp(out), _ip(status)))
        return out, status

    def likes2d(self, d_hist, d_likehist, B, F, rx, ry, corr, winw, flags, mbc):
        out = self.alloc(B * F * F * 8)
        rx, ry, corr, winw, flags = _f64arr(rx), _f64arr(ry), _f64arr(corr), _i32arr(winw), _i32arr(flags)
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_likes2d(self.h, B, F, d_hist.ptr, d_likehist.ptr, _dp(rx), _dp(ry), _dp(corr),
                                        _ip(winw), _ip(flags), int(mbc), out.ptr, _ip(status)))
        return out, status

    # ---- stand-alone convolutions / likelihood statistics
    def circ_convolve(self, a, b):
        """irfft(rfft(a) * rfft(b)) of two equal-shape real arrays (1-D or 2-D) through rocFFT."""
        a, b = _f64arr(a), _f64arr(b)
        if a.shape != b.shape or a.ndim not in (1, 2):
            raise ValueError("circ_convolve needs two arrays of one shape, one- or two-dimensional")
        n0, n1 = (1, a.shape[0]) if a.ndim == 1 else a.shape
        out = np.empty_like(a)
        self._check(self.lib.gd_circ_convolve(self.h, int(n0), int(n1), _dp(a), _dp(b), _dp(out)))
        return out

    def convolve1d_direct(self, x, y):
        x, y = _f64arr(x), _f64arr(y)
        out = np.empty(x.size + y.size - 1)
        self._check(self.lib.gd_convolve1d_direct(self.h, _dp(x), x.size, _dp(y), y.size, _dp(out)))
        return out

    def autoconvolve(self, s, n, normalize=True, x=None, col=-1, mean=0.0, use_weights=False):
        """autoConvolve of a host vector ``x`` or of (column - mean) * weights; ``s`` = nearestFFTnumber(2 N)."""
        out = np.empty(int(n))
        xx = None if x is None else _f64arr(x)
        self._check(self.lib.gd_autoconvolve(self.h, int(col), float(mean), int(bool(use_weights)),
                                             None if xx is None else _dp(xx), 0 if xx is None else xx.size, int(s), int(n),
                                             int(bool(normalize)), _dp(out)))
        return out

    def like_stats(self, col):
        out = np.zeros(8)
        self._check(self.lib.gd_like_stats(self.h, int(col), _dp(out)))
        return dict(min=out[0], max=out[1], norm=out[2], sum_wl=out[3], sum_wl2=out[4], sum_w_exp_plus=out[5],
                    sum_w_exp_minus=out[6], argmin=int(out[7]))

    # ---- one native entry for a batch of pairs (getdist_amd/batch2d.py packs the arguments)
    def batch2d_grid_sizes(self, settings, n, corr, pairs32):
        F = np.zeros(len(pairs32), dtype=np.int32)
        rc = self.lib.gd_batch2d_grid_sizes(C.byref(settings), int(n), _dp(corr), _ip(pairs32), len(pairs32), _ip(F))
        if rc != 0:
            raise GdhipError(rc, "gd_batch2d_grid_sizes: bad argument")
        return F

    def density2d_batch(self, twin, settings, params, n, corr, cov, lag_probe, pairs32, exchange, grids, status, meta, levels,
                        level_status):
        """gd_density2d_batch; returns the copy-stream tokens (this context's, the twin's)."""
        tokens = np.full(2, -1, dtype=np.int32)
        self._check(self.lib.gd_density2d_batch(
            self.h, None if twin is None else twin.h, C.byref(settings), C.cast(params, _p), int(n), _dp(corr), _dp(cov),
            None if lag_probe is None else _dp(lag_probe), _ip(pairs32), len(pairs32),
            None if exchange is None else C.cast(exchange, _p), None, grids.ctypes.data, int(grids.size),
            status.ctypes.data_as(_pi32), _dp(meta), None if levels is None else _dp(levels),
            None if level_status is None else _ip(level_status), _ip(tokens)))
        return int(tokens[0]), int(tokens[1])

    # ---- RCCL communicator of a multi-GPU job (one process per GPU)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        rc = load_library().gd_comm_unique_id(buf)
        if rc != 0:
            raise GdhipError(rc, "gd_comm_unique_id failed (librccl.so missing?)")
        return buf.raw

    def comm_init(self, world, rank, id128):
        self._check(self.lib.gd_comm_init(self.h, int(world), int(rank), C.create_string_buffer(bytes(id128), 128)))
        self.comm_world, self.comm_rank = int(world), int(rank)

    def comm_allgather(self, vec):
        """(world, len(vec)) array: every rank's vector, through ncclAllGather on the context's stream."""
        v = _f64arr(vec).ravel()
        out = np.empty((self.comm_world, v.size))
        self._check(self.lib.gd_comm_allgather(self.h, _dp(v), v.size, _dp(out)))
        return out

    def comm_allreduce_sum(self, vec):
        v = _f64arr(vec).ravel().copy()
        self._check(self.lib.gd_comm_allreduce_sum(self.h, _dp(v), v.size))
        return v

    def comm_destroy(self):
        self._check(self.lib.gd_comm_destroy(self.h))
        self.comm_world = 0

    def comm_info(self):
        """(world, rank) of the context's communicator as the LIBRARY sees it; (0, 0) when there is none (never made, or
        aborted and dropped after a timeout)."""
        wd, rk = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.gd_comm_info(self.h, C.byref(wd), C.byref(rk)))
        return int(wd.value), int(rk.value)

    def comm_abandon(self):
        """The one call that may be made from another thread while a gd_comm_* call of this context has not returned: a
        gd_comm_init that comes back after its caller gave up installs nothing."""
        if self.h is not None:
            self.lib.gd_comm_abandon(self.h)

    def density1d_batch(self, settings, params, n, cols32, want_hist=False):
        """gd_density1d_batch: (P[B, F], hist[B, F] or None, meta[B, 9]) of the listed columns."""
        cols32 = _i32arr(cols32)
        B, F = len(cols32), int(settings.fine_bins)
        P = np.zeros((B, F))
        hist = np.zeros((B, F)) if want_hist else None
        meta = np.zeros((B, 9))
        self._check(self.lib.gd_density1d_batch(self.h, C.byref(settings), C.cast(params, _p), int(n), _ip(cols32), B, _dp(P),
                                                None if hist is None else _dp(hist), _dp(meta)))
        return P, hist, meta

    def batch2d_finish(self):
        self._check(self.lib.gd_batch2d_finish(self.h))

    def batch2d_invalidate(self):
        self._check(self.lib.gd_batch2d_invalidate(self.h))

    def batch2d_exchanges(self):
        """N_eff collectives gd_density2d_batch has entered on this context so far (also by calls that failed later)."""
        c = C.c_int64(0)
        self._check(self.lib.gd_batch2d_exchanges(self.h, C.byref(c)))
        return int(c.value)

    def kopt2d(self, d_hist, B, F, neff, do_corr, fallback_t, corr):
        """B x 12: {t*, psi_02, psi_20, psi_11, psi_00, psi_13, psi_31, status, hx, hy, corr, get_h status}"""
        neff, do_corr, fallback_t, corr = _f64arr(neff), _i32arr(do_corr), _f64arr(fallback_t), _f64arr(corr)
        out = np.zeros((B, 12))
        self._check(self.lib.gd_kopt2d(self.h, B, F, d_hist.ptr if isinstance(d_hist, DevBuf) else d_hist, _dp(neff),
                                       _ip(do_corr), _dp(fallback_t), _dp(corr), _dp(out)))
        return out

    def get_h(self, psi, neff, corr, do_corr):
        """B x 4 {hx, hy, corr, status} from the functionals psi (B x 6): KernelOptimizer2D.get_h on the device."""
        psi, neff, corr, do_corr = _f64arr(psi).reshape(-1, 6), _f64arr(neff), _f64arr(corr), _i32arr(do_corr)
        out = np.zeros((len(psi), 4))
        self._check(self.lib.gd_get_h(self.h, len(psi), _dp(psi), _dp(neff), _dp(corr), _ip(do_corr), _dp(out)))
        return out

    def density2d(self, d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc, out=None):
        out = out or self.alloc(B * F * F * 8)
        rx, ry, corr, winw, flags = _f64arr(rx), _f64arr(ry), _f64arr(corr), _i32arr(winw), _i32arr(flags)
        status = np.zeros(B, dtype=np.int32)
        self._check(self.lib.gd_density2d(self.h, B, F, d_hist.ptr if isinstance(d_hist, DevBuf) else d_hist, _dp(rx),
                                          _dp(ry), _dp(corr), _ip(winw), _ip(flags), int(bco), int(mbc), out.ptr,
                                          _ip(status)))
        return out, status
