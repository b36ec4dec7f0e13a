"""
CPU tests of the host-side scalar solvers kept in the product (getdist_amd/mcsamples.py): given the SAME inputs
as the oracle they must return the oracle's result bit-for-bit -- this is the deterministic half of the
solver-path parity argument (the chaotic half is documented in DESIGN.md).
"""

import numpy as np
from scipy import fftpack

from getdist_amd import mcsamples as hm
from oracle import kde_oracle as ko


def test_isj_solve_matches_oracle(zoo):
    fx = zoo["shapes"]
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for j in range(orc.n):
        d = orc.density_1d(j)
        neff = orc.pars[j].N_eff_kde
        a = fftpack.dct(d["bins"] / np.sum(d["bins"]))
        assert hm._isj_solve(a, neff) == ko.isj_bandwidth_binned(d["bins"], neff)


def test_get_h_matches_oracle(zoo):
    fx = zoo["block10_weighted"]
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    n_tnc = 0
    for (a, b) in fx["pairs"]:
        d = orc.density_2d(a, b)
        for corr, do_corr in ((0.0, True), (0.15, True), (-0.4, True), (0.3, False)):
            tr = {}
            opt = ko.Optimizer2D(d["histbins"], 4000.0, corr, do_correlation=do_corr, fallback_t=1e-4, trace=tr)
            want = opt.get_h()
            psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr.get("p_00", np.nan), tr.get("p_13", np.nan),
                   tr.get("p_31", np.nan))
            got = hm._get_h(psi, 4000.0, corr, do_corr)
            assert tuple(map(float, got)) == tuple(map(float, want)), (a, b, corr, got, want)
            n_tnc += do_corr
    assert n_tnc > 0


def test_cov_to_corr_and_bounds():
    c = np.array([[4.0, 1.0], [1.0, 9.0]])
    assert np.allclose(hm.covToCorr(c), ko.cov_to_corr(c))
    b = hm.ParamBounds()
    b.setRange("x", (0, None))
    b.setRange("phi", (0, 6.28, "periodic"))
    assert b.getLower("x") == 0.0 and b.getUpper("x") is None and "phi" in b.periodic
