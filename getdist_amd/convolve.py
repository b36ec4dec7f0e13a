"""
Device-backed forms of getdist/convolve.py's public functions (convolve.py:5-478): same names, arguments, modes and
result shapes; the transforms and products run on the GPU through libgdhip.so (rocFFT for the transform pair, a
direct-sum kernel where the reference calls np.convolve), the padding, the centring roll and the mode slices are the
reference's.  The batched density pipeline does NOT come through here (it keeps its frames on the device,
csrc/density1d.hip / density2d.hip); these are the entry points for code that calls ``getdist.convolve`` directly.

A context is needed: the functions take ``ctx=`` (a ``getdist_amd._lib.Context`` or anything with the same three
methods) and otherwise use a lazily created context on device 0.  There is no CPU fallback.

``nearestFFTnumber``: the reference picks its zero-padded FFT sizes from a literal table; the same numbers are
generated here from their rule (2^a 3^b 5^c, c <= 1, for the row ranges below, plus the table's two stragglers).
"""

import numpy as np

_ROWS = {(0, 0): (1, 29), (0, 1): (1, 28), (1, 0): (1, 29), (1, 1): (5, 27), (2, 0): (4, 27), (2, 1): (4, 25), (3, 0): (4, 26)}
fastFFT = np.array(sorted([7 * 2**25, 81 * 2**24] + [2**a * 3**b * 5**c for (b, c), (lo, hi) in _ROWS.items()
                                                      for a in range(lo, hi + 1)]), dtype=np.int64)


def nearestFFTnumber(x):
    return np.maximum(x, fastFFT[np.searchsorted(fastFFT, x)])


_default_ctx = None


def set_context(ctx):
    """Use ``ctx`` for calls that do not pass one (None: create a context on device 0 at the next call)."""
    global _default_ctx
    _default_ctx = ctx


def _ctx(ctx):
    global _default_ctx
    if ctx is not None:
        return ctx
    if _default_ctx is None:
        from ._lib import Context

        _default_ctx = Context(0)
    return _default_ctx


def _pad(a, shape):
    out = np.zeros(shape, dtype=np.float64)
    out[tuple(slice(0, n) for n in a.shape)] = a
    return out


def convolveFFT(x, y, mode="same", yfft=None, xfft=None, largest_size=0, cache=None, cache_args=(1, 2), ctx=None):
    """convolve.py:371-401.  The FFT caches of the reference have no meaning here (operands are transformed on the
    device every time); ``cache`` / ``xfft`` / ``yfft`` are accepted and ignored."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    size = x.size + y.size - 1
    fsize = int(nearestFFTnumber(np.maximum(largest_size, size)))
    res = _ctx(ctx).circ_convolve(_pad(x, (fsize,)), _pad(y, (fsize,)))[0:size]
    if mode == "same":
        return res[(y.size - 1) // 2:(y.size - 1) // 2 + x.size]
    elif mode == "full":
        return res
    elif mode == "valid":
        return res[y.size - 1:x.size]
    raise ValueError("unknown convolution mode %s" % mode)


def _np_convolve_slice(full, nx, ny, mode):
    """The slice of the full convolution that np.convolve(x, y, mode) returns (numpy/_core/numeric.py: 'same' is
    centred on the longer operand, 'valid' has max - min + 1 points)."""
    lo, hi = min(nx, ny), max(nx, ny)
    if mode == "full":
        return full
    if mode == "same":
        start = (lo - 1) // 2
        return full[start:start + hi]
    if mode == "valid":
        return full[lo - 1:hi]
    raise ValueError("unknown convolution mode %s" % mode)


def convolve1D(x, y, mode, largest_size=0, cache=None, cache_args=(1, 2), ctx=None):
    """convolve.py:196-202"""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if mode == "periodic":
        return convolve1D_periodic(x, y, cache, cache_args, ctx=ctx)
    if min(x.shape[0], y.shape[0]) > 1000:
        return convolveFFT(x, y, mode, largest_size=largest_size, ctx=ctx)
    return _np_convolve_slice(_ctx(ctx).convolve1d_direct(x, y), x.size, y.size, mode)


def convolve1D_periodic(x, y, cache=None, cache_args=(1, 2), ctx=None):
    """convolve.py:326-367: circular convolution of the folded grid (last bin added to the first), result re-extended"""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    x_circ = x[:-1].copy()
    x_circ[0] += x[-1]
    N, M = x_circ.shape[0], y.shape[0]
    hpad = np.zeros(N)
    hpad[:M] = y
    hpad = np.roll(hpad, -(M // 2))
    res = _ctx(ctx).circ_convolve(x_circ, hpad)
    return np.append(res, res[0])


def _centered(arr, newsize):
    startind = (np.array(arr.shape) - newsize) // 2
    endind = startind + newsize
    return arr[tuple(slice(startind[k], endind[k]) for k in range(len(endind)))]


def convolveFFTn(in1, in2, mode="same", largest_size=0, cache=None, yfft=None, xfft=None, cache_args=(1, 2), ctx=None):
    """convolve.py:405-436 (two-dimensional operands)"""
    in1, in2 = np.asarray(in1, dtype=np.float64), np.asarray(in2, dtype=np.float64)
    if in1.ndim != 2 or in2.ndim != 2:
        raise ValueError("convolveFFTn: two-dimensional arrays")
    s1, s2 = np.array(in1.shape), np.array(in2.shape)
    size = s1 + s2 - 1
    fsize = tuple(int(v) for v in nearestFFTnumber(np.maximum(largest_size, size)))
    ret = _ctx(ctx).circ_convolve(_pad(in1, fsize), _pad(in2, fsize))[tuple(slice(0, int(sz)) for sz in size)]
    if mode == "full":
        return ret
    elif mode == "same":
        return _centered(ret, s1)
    elif mode == "valid":
        return _centered(ret, s1 - s2 + 1)
    raise ValueError("unknown convolution mode %s" % mode)


def convolve2D(x, y, mode, largest_size=0, cache=None, cache_args=(1, 2), ctx=None):
    """convolve.py:205-212"""
    if mode in ("periodic", "periodic_both"):
        return convolve2D_periodic(x, y, cache, cache_args, periodic_x=True, periodic_y=True, ctx=ctx)
    elif mode == "periodic_x":
        return convolve2D_periodic(x, y, cache, cache_args, periodic_x=True, periodic_y=False, ctx=ctx)
    elif mode == "periodic_y":
        return convolve2D_periodic(x, y, cache, cache_args, periodic_x=False, periodic_y=True, ctx=ctx)
    return convolveFFTn(x, y, mode, largest_size, ctx=ctx)


def convolve2D_periodic(x, y, cache=None, cache_args=(1, 2), periodic_x=True, periodic_y=True, ctx=None):
    """convolve.py:215-323: fold the periodic axes, circular convolution with the centred kernel, extend again"""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if x.ndim != 2 or y.ndim != 2:
        raise ValueError("convolve2D_periodic requires 2D arrays")
    if not periodic_x and not periodic_y:
        return convolveFFTn(x, y, "same", ctx=ctx)
    ny, nx = x.shape
    ky, kx = y.shape
    if periodic_x and periodic_y:  # the reference's order of additions (the corner bin receives three terms)
        x_circ = x[:-1, :-1].copy()
        x_circ[0, :] += x[-1, :-1]
        x_circ[:, 0] += x[:-1, -1]
        x_circ[0, 0] += x[-1, -1]
    elif periodic_x:
        x_circ = x[:, :-1].copy()
        x_circ[:, 0] += x[:, -1]
    else:
        x_circ = x[:-1, :].copy()
        x_circ[0, :] += x[-1, :]
    N_y, N_x = x_circ.shape
    hpad = np.zeros((N_y, N_x))
    hpad[:ky, :kx] = y
    hpad = np.roll(np.roll(hpad, -(ky // 2), axis=0), -(kx // 2), axis=1)
    result = _ctx(ctx).circ_convolve(np.ascontiguousarray(x_circ), hpad)
    final = np.empty((ny, nx))
    final[:N_y, :N_x] = result
    if periodic_x:
        final[:N_y, -1] = result[:, 0]
    if periodic_y:
        final[-1, :N_x] = result[0, :]
    if periodic_x and periodic_y:
        final[-1, -1] = result[0, 0]
    return final


def autoConvolve(x, n=None, normalize=True, ctx=None):
    """convolve.py:458-478: result[k] = sum_i x_i x_{i+k} (k < n), divided by the number of terms if ``normalize``"""
    x = np.asarray(x, dtype=np.float64)
    s = int(nearestFFTnumber(2 * x.size))
    n = n or x.size
    return _ctx(ctx).autoconvolve(s, n, normalize, x=x)


def autoCorrelation(x, n=None, normalized=True, start_index=0, ctx=None):
    """convolve.py:446-455"""
    x = np.asarray(x, dtype=np.float64)
    result = autoConvolve(x - x.mean(), n, normalize=True, ctx=ctx)
    if normalized:
        result /= result[0]
    return result[start_index:]
