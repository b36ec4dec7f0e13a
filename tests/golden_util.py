"""Helpers shared by the golden-vector tests (oracle vs goldens on CPU; HIP path vs goldens on GPU)."""

import os
import zlib

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PAR_ATTS = ("param_min", "param_max", "range_min", "range_max", "sigma_range", "err", "mean", "has_limits_bot",
            "has_limits_top", "N_eff_kde", "kde_h")


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, "fixture_%s.npz" % name))


def kwkey(kw):
    return ",".join("%s=%s" % (k, kw[k]) for k in sorted(kw)) or "default"


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def relerr(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    den = np.max(np.abs(b)) or 1.0
    return float(np.max(np.abs(a - b)) / den)


def check_grid_2d(gold, key, P, tol):
    """Compare a computed P[y,x] grid with whichever form the golden file holds."""
    st = int(gold[key + "/stride"])
    if key + "/P" in gold.files:
        assert relerr(P, gold[key + "/P"]) <= tol, key
    else:
        assert relerr(P[::st, ::st], gold[key + "/Pstrided"]) <= tol, key
    assert abs(np.sum(P) - float(gold[key + "/Psum"])) <= tol * float(gold[key + "/Psum"]) * 10, key


def meanlikes_cases(zoo, g):
    """Yields (case, kw1, kw2, fixture, loglikes) for the mean-likelihood goldens (tests/golden/meanlikes.npz)."""
    from oracle.fixtures import MEANLIKES_CASES, loglikes_for

    for nm, kws in MEANLIKES_CASES:
        fx = zoo[nm]
        ll = loglikes_for(fx["samples"])
        for kw in kws:
            yield ("%s/%s" % (nm, kwkey(kw)), {k: v for k, v in kw.items() if k != "fine_bins_2D"},
                   {k: v for k, v in kw.items() if k != "fine_bins"}, fx, ll)


def likes_outliers(a, ref, tol=1e-6):
    """
    Number of pixels where two mean-likelihood grids differ by more than tol.  The reference's FFT route leaves a
    handful of noise-decided pixels in its own output (its `bin2Dlikes > 0` mask follows the sign of ~1e-14 rounding
    noise where the likelihood weight is tiny; DESIGN.md "mean likelihoods"), so grids evaluated without that noise
    agree with the reference everywhere except at those isolated pixels.
    """
    return int(np.sum(np.abs(np.asarray(a) - np.asarray(ref)) > tol))


MAX_LIKES_OUTLIERS = 8  # measured: at most 5 per grid in the fixture zoo (oracle/validate_against_reference.py)
