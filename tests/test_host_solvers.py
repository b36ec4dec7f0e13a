"""
CPU tests of small host-side helpers of the product (getdist_amd/mcsamples.py, convolve.py).  The scalar solvers
themselves run on the device; their CPU pinning against scipy is in tests/test_native_solvers.py.
"""

import numpy as np

from getdist_amd import mcsamples as hm
from oracle import kde_oracle as ko


def test_cov_to_corr_and_bounds():
    c = np.array([[4.0, 1.0], [1.0, 9.0]])
    assert np.allclose(hm.covToCorr(c), ko.cov_to_corr(c))
    b = hm.ParamBounds()
    b.setRange("x", (0, None))
    b.setRange("phi", (0, 6.28, "periodic"))
    assert b.getLower("x") == 0.0 and b.getUpper("x") is None and "phi" in b.periodic


def test_nearest_fft_number_api():
    import golden_util as gu
    from getdist_amd.convolve import nearestFFTnumber

    g = np.load(gu.GOLDEN_DIR + "/fftnumbers.npz")
    assert np.array_equal(nearestFFTnumber(g["x"]), g["y"])
