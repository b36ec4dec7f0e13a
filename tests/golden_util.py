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


def reference_unit_test_checks(factory=None):
    """The reference's OWN unit tests for this path (getdist/tests/getdist_test.py: testFileLoadPlot, testLimits) on its
    own inputs (tests/golden/reference_unit_tests.npz, made by make_golden.py --reference-unit-tests): the asserted
    numbers to the reference's assertAlmostEqual places, and the reference's actual outputs to 1e-9."""
    import numpy as np

    from getdist_amd.mcsamples import MCSamples

    g = np.load(GOLDEN_DIR + "/reference_unit_tests.npz")
    kw = {} if factory is None else dict(_context_factory=factory)
    off = g["fileload/chain_offsets"]
    cut = lambda a: [a[lo:hi] for lo, hi in zip(off[:-1], off[1:])]  # noqa: E731
    mc = MCSamples(samples=cut(g["fileload/samples"]), weights=cut(g["fileload/weights"]),
                   loglikes=cut(g["fileload/loglikes"]), names=["x", "y"], settings={"ignore_rows": 0.1}, **kw)
    assert mc.numrows == int(g["fileload/numrows_after_burn"])
    mc.getConvergeTests(0.95)
    assert round(abs(mc.GelmanRubin - float(g["fileload/asserted"])), 4) == 0  # assertAlmostEqual(..., 4)
    assert abs(mc.GelmanRubin - float(g["fileload/GelmanRubin"])) < 1e-9 * float(g["fileload/GelmanRubin"])
    names = [str(n) for n in g["limits/names"]]
    ranges = {n: (None if np.isnan(lo) else float(lo), None if np.isnan(hi) else float(hi))
              for n, lo, hi in zip(names, g["limits/range_lo"], g["limits/range_hi"])}
    tl = MCSamples(samples=g["limits/samples"], names=names, ranges=ranges, **kw)
    lims = tl.getMargeStats().parWithName("x").limits
    for k in (0, 1):
        assert round(abs(lims[k].lower - float(g["limits/asserted"][k])), 3) == 0  # assertAlmostEqual(..., 3)
        assert abs(lims[k].lower - float(g["limits/x_lower"][k])) < 1e-6
    assert bool(lims[2].onetail_lower) == bool(g["limits/x_onetail_lower_2"])
