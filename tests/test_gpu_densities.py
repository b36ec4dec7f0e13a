"""
GPU parity tests of the full 1D/2D KDE path (MCSamples API -> C ABI -> HIP kernels) against the oracle on the
seeded fixture zoo and against the committed reference outputs (tests/golden).

Tolerances (BASELINE.json north_star): bin indices bit-exact (tests/test_gpu_primitives.py), density grids within
1e-6 of the grid maximum (P is max-normalised, so absolute == relative-to-max), bandwidths 1e-6 relative (they pass
through scipy solvers with loose stopping rules).
"""

import numpy as np
import pytest

import golden_util as gu
from oracle import kde_oracle as ko

pytestmark = pytest.mark.gpu

TOL_GRID = 1e-6
FIXTURES = ["c1_100k", "c1_bounded", "block10_weighted", "block50", "shapes", "shapes_intweights"]


def make(fx):
    from getdist_amd.mcsamples import MCSamples

    return MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])


@pytest.mark.parametrize("name", FIXTURES)
def test_stats_and_ranges(zoo, name):
    fx = zoo[name]
    g = gu.load(name)
    mc = make(fx)
    assert gu.relerr(mc.means, g["means"]) < 1e-12
    assert gu.relerr(mc.vars, g["vars"]) < 1e-12
    assert gu.relerr(mc.fullcov, g["cov"]) < 1e-11
    assert gu.relerr(mc.getCorrelationMatrix(), g["corr"]) < 1e-11
    fracs = g["quantile_fracs"]
    for j in range(mc.n):
        q = mc.confidence(j, fracs)
        if fx["weights"] is None or name == "shapes_intweights":
            assert np.array_equal(q, g["quantiles"][j])
        else:
            assert np.mean(q == g["quantiles"][j]) >= 0.8


@pytest.mark.parametrize("name", FIXTURES + ["periodic"])
def test_density_1d(zoo, name):
    fx = zoo[name]
    g = gu.load(name)
    mc = make(fx)
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for kw in fx["kw1"]:
        dens = mc.get1DDensities(**kw)
        for j, nm in enumerate(fx["names"]):
            d = dens[j]
            o = orc.density_1d(j, **kw)
            key = "p1d/%s/%s" % (nm, gu.kwkey(kw))
            assert np.max(np.abs(d.P - o["P"])) < TOL_GRID, (key, "vs oracle")
            assert np.max(np.abs(d.P - g[key + "/P"])) < TOL_GRID, (key, "vs golden")
            assert np.array_equal([d.x[0], d.x[-1]], g[key + "/x0x1"]) or \
                np.allclose([d.x[0], d.x[-1]], g[key + "/x0x1"], rtol=1e-13, atol=0), key
            if not kw:
                par = mc.paramNames.parWithName(nm)
                got = np.array([float(getattr(par, a)) for a in gu.PAR_ATTS])
                want = g["par/%s" % nm]
                assert np.array_equal(got[7:9], want[7:9]), (nm, "limit flags")
                assert gu.relerr(got[:7], want[:7]) < 1e-11, (nm, got, want)
                assert abs(got[9] - want[9]) <= 1e-9 * want[9], (nm, "N_eff", got[9], want[9])
                assert abs(got[10] - want[10]) <= 1e-6 * want[10], (nm, "kde_h", got[10], want[10])


@pytest.mark.parametrize("name", FIXTURES)
def test_density_2d(zoo, name):
    fx = zoo[name]
    g = gu.load(name)
    mc = make(fx)
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for kw in fx["kw2"]:
        dens = mc.get2DDensities(fx["pairs"], get_density=False, **kw)
        for (a, b), d in zip(fx["pairs"], dens):
            key = "p2d/%s/%s/%s" % (fx["names"][a], fx["names"][b], gu.kwkey(kw))
            tr = {}
            o = orc.density_2d(a, b, trace=tr, **kw)
            assert d.P.shape == o["P"].shape, key
            if key + "/hxhyc" in g.files:
                assert gu.relerr(d.bandwidth, g[key + "/hxhyc"]) < 1e-6, (key, d.bandwidth, g[key + "/hxhyc"])
            assert np.max(np.abs(d.P - o["P"])) < TOL_GRID, (key, "vs oracle", np.max(np.abs(d.P - o["P"])))
            gu.check_grid_2d(g, key, d.P, TOL_GRID)
            assert gu.relerr(d.contours, g[key + "/contours"]) < 1e-5, key
            assert np.allclose([d.x[0], d.x[-1], d.y[0], d.y[-1]], g[key + "/xy"], rtol=1e-12, atol=0), key


def test_single_pair_api_and_cache(zoo):
    fx = zoo["c1_bounded"]
    mc = make(fx)
    d1 = mc.get1DDensity("a")
    assert mc.get1DDensity("a") is d1  # cached per name when no kwargs (mcsamples.py:1509-1513)
    assert mc.get1DDensity("a", fine_bins=512) is not d1
    d2 = mc.get2DDensity("a", "d")
    assert d2.P.shape == (256, 256) and np.max(d2.P) == 1.0
    assert mc.get2DDensity("nope", "d") is None
    dn = mc.get2DDensity("a", "d", normalized=True)
    assert abs(dn.norm_integral() - 1) < 1e-12


def test_gelman_rubin_golden():
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    g = np.load(gu.GOLDEN_DIR + "/convergence.npz")
    samples, weights, names, offsets = synth.config_c4(nchains=4, N=20000, n=8)
    chains = [samples[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    ws = [weights[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    mc = MCSamples(samples=chains, weights=ws, names=names)
    D = mc.getGelmanRubinEigenvalues()
    assert gu.relerr(D, g["gr_eigenvalues"]) < 1e-9
    assert abs(mc.getGelmanRubin() - float(g["gr"])) < 1e-9 * float(g["gr"])
    assert gu.relerr(mc.getMeanVarTest(), g["meanvar"]) < 1e-9
