"""
The native route of MCSamples.get1DDensities: one call of gd_density1d_batch (csrc/batch1d.hpp) -- bin edges, binning, the
ISJ bandwidth with its scalar tail (mcsamples.py:1256-1283), the smoothing scale and the window, convolution, boundary
and bias correction, normalisation (mcsamples.py:1500-1686) -- for all listed parameters.  What stays here is what the
reference does around those numbers: warnings, exceptions, the parameters' cached kde_h / N_eff_kde.
"""

import ctypes as C
import logging

import numpy as np

from .batch2d import NEED_NEFF, pack_params


class Density1DSettings(C.Structure):
    """gd_density1d_settings"""

    _fields_ = [("fine_bins", C.c_int32), ("num_bins", C.c_int32), ("boundary_correction_order", C.c_int32),
                ("mult_bias_correction_order", C.c_int32), ("smooth_scale_1D", C.c_double), ("norm", C.c_double),
                ("sum_w2", C.c_double), ("uncorrelated_sampler", C.c_int32), ("raise_on_bandwidth_errors", C.c_int32)]


def run(mc, js, fine_bins, num_bins, smooth_scale_1D, bco, mbc, want_hist):
    """(P[B, F], hist[B, F] or None, meta[B, 9]) for the columns ``js`` (their parameters initialised by the caller)."""
    from . import mcsamples as M
    from ._lib import GdhipError

    names = mc.paramNames.names
    s = Density1DSettings()
    s.fine_bins, s.num_bins, s.boundary_correction_order, s.mult_bias_correction_order = fine_bins, num_bins, bco, mbc
    s.smooth_scale_1D = smooth_scale_1D
    s.norm = float(mc.norm)
    s.sum_w2 = float(mc._sum_w2)
    s.uncorrelated_sampler = int(mc.sampler in ("nested", "uncorrelated"))
    s.raise_on_bandwidth_errors = int(bool(mc.raise_on_bandwidth_errors))
    cols32 = np.asarray(js, dtype=np.int32)
    uniq = list(dict.fromkeys(js))
    for attempt in (0, 1):
        params = pack_params(mc, uniq)
        try:
            P, hist, meta = mc.ctx.density1d_batch(s, params, mc.n, cols32, want_hist=want_hist)
            break
        except GdhipError as e:
            if e.code == NEED_NEFF and attempt == 0:
                mc._neff_batch(uniq)  # getCorrelationLength's long route; cached on the parameters for the second call
                continue
            msg = str(e).split(": ", 1)[-1]
            if e.code == -5:
                raise M.BandwidthError(_as_python_prints(_with_names(msg, names)))
            if "Parameter range is <= 0" in msg:
                raise M.MCSamplesError("Parameter range is <= 0: " + names[int(msg.rsplit(" ", 1)[-1])].name)
            if e.code == -1:
                raise M.SettingError(msg)
            raise
    for j in uniq:
        if names[j].N_eff_kde is None and not np.isnan(params[j].neff):
            names[j].N_eff_kde = float(params[j].neff)
    bits = meta[:, 5].astype(np.int64)
    for b, j in enumerate(js):
        par = names[j]
        if bits[b] & 1:
            logging.warning("1D auto bandwidth failed. Using fallback: zero f in _bandwidth_fixed_point (non-convergence)")
        if bits[b] & 2:  # the reference's message (mcsamples.py:1262): the solver's own width, N_eff, the fallback width
            h_isj = None if np.isnan(meta[b, 8]) else float(meta[b, 8])
            logging.warning(f"auto bandwidth for {par.name} very small or failed (h={h_isj},N_eff={float(meta[b, 6])}). "
                            f"Using fallback (h={float(meta[b, 2])})")
        if bits[b] & 4:
            logging.warning("fine_bins not large enough to well sample smoothing scale - " + par.name)
        if smooth_scale_1D <= 0:
            par.kde_h = float(meta[b, 2])
    if np.any(meta[:, 7] != 0):
        raise M.DensitiesError("no samples in bin")
    return P, hist, meta


def _as_python_prints(msg):
    """the library writes its doubles with 17 significant digits; the reference's f-string shows repr(float)"""
    import re

    return re.sub(r"=(-?\d[0-9.e+-]*)", lambda m: "=" + repr(float(m.group(1))), msg)


def _with_names(msg, names):
    """'... for column 7 ...' -> the parameter's name, as the reference's message has it"""
    import re

    return re.sub(r"column (\d+)", lambda m: names[int(m.group(1))].name, msg)
