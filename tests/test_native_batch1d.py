"""
CPU tests of gd_density1d_batch's HOST side (getdist_amd/csrc/batch1d.hpp), compiled with g++ and driven against the numpy
context double: the 1D densities, bandwidths and cached parameter numbers of MCSamples.get1DDensities through the native
entry's code must equal those of the Python-planned sequence bit for bit.  (On the GPU the same comparison runs against the
real kernels: tests/test_gpu_densities.py.)
"""

import ctypes as C
import logging

import numpy as np
import pytest

import native_batch_util as nb
from test_native_batch import make


def same1d(native, plain):
    assert len(native) == len(plain)
    for k, (a, b) in enumerate(zip(native, plain)):
        assert np.array_equal(a.x, b.x), k
        assert np.array_equal(a.P, b.P), (k, float(np.max(np.abs(a.P - b.P))))
        assert (a.likes is None) == (b.likes is None) and (a.likes is None or np.array_equal(a.likes, b.likes)), k
        assert a.view_ranges == b.view_ranges


@pytest.mark.parametrize("name", ["block10_weighted", "c1_bounded", "periodic"])
def test_native_1d_route_equals_python_route(zoo, name):
    fx = zoo[name]
    ref = make(fx, nb.PlainContext)
    plain = ref.get1DDensities()
    nb.CALLS.clear()
    mc = make(fx, nb.HarnessContext)
    native = mc.get1DDensities()
    same1d(native, plain)
    ops = [c[0] for c in nb.CALLS]
    assert ops.count("hist1d_dev") == 1 and ops.count("isj1d_dev") == 1 and ops.count("density1d_dev") == 1
    assert [p.kde_h for p in mc.paramNames.names] == [p.kde_h for p in ref.paramNames.names]
    assert [p.N_eff_kde for p in mc.paramNames.names] == [p.N_eff_kde for p in ref.paramNames.names]
    # the densities are cached per name, and a second call finds the effective sample numbers known
    nb.CALLS.clear()
    same1d(mc.get1DDensities(), plain)
    assert not any(c[0] in ("autocov_lags_batch", "kde_lag_sums_batch") for c in nb.CALLS)


@pytest.mark.parametrize("kw", [dict(smooth_scale_1D=0.3), dict(smooth_scale_1D=1.7, num_bins=60), dict(smooth_scale_1D=-2.0),
                                dict(boundary_correction_order=0), dict(boundary_correction_order=2),
                                dict(mult_bias_correction_order=0), dict(mult_bias_correction_order=2), dict(fine_bins=512),
                                dict(boundary_correction_order=2, mult_bias_correction_order=0)])
def test_native_1d_route_settings(zoo, kw):
    fx = zoo["c1_bounded"]
    js = [3, 0, 2]
    plain = make(fx, nb.PlainContext).get1DDensities(js, **kw)
    native = make(fx, nb.HarnessContext).get1DDensities(js, **kw)
    same1d(native, plain)


def test_native_1d_route_mean_likelihoods(zoo):
    fx = zoo["block10_weighted"]
    rng = np.random.default_rng(4)
    loglikes = rng.chisquare(4, size=len(fx["samples"])) / 2

    def mk(factory):
        from getdist_amd.mcsamples import MCSamples

        return MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], loglikes=loglikes,
                         _context_factory=factory)

    same1d(mk(nb.HarnessContext).get1DDensities([0, 4, 5], meanlikes=True), mk(nb.PlainContext).get1DDensities([0, 4, 5], meanlikes=True))


def test_native_1d_errors_and_warnings(zoo, caplog):
    from getdist_amd.mcsamples import BandwidthError, SettingError

    fx = zoo["c1_bounded"]
    mc = make(fx, nb.HarnessContext)
    with pytest.raises(SettingError):
        mc.get1DDensities([0], boundary_correction_order=3)
    # a fine grid far too coarse for the smoothing scale: the reference's warning, same density as the planned route
    with caplog.at_level(logging.WARNING):
        native = mc.get1DDensities([1], smooth_scale_1D=0.01)
    assert any("fine_bins not large enough" in r.message for r in caplog.records)
    same1d(native, make(fx, nb.PlainContext).get1DDensities([1], smooth_scale_1D=0.01))
    # the rule-of-thumb fallback raises when asked to (mcsamples.py:1270-1275): a width the check calls "very small"
    lib = nb.harness()
    from getdist_amd.batch1d import Density1DSettings
    from getdist_amd.batch2d import ParamState

    s = Density1DSettings(fine_bins=1024, num_bins=100, boundary_correction_order=1, mult_bias_correction_order=1,
                          smooth_scale_1D=-1.0, norm=1.0, sum_w2=1.0, uncorrelated_sampler=0, raise_on_bandwidth_errors=1)
    p = ParamState(range_min=0.0, range_max=1.0, param_min=0.0, param_max=1.0, sigma_range=0.2, err=0.2, neff=1000.0)
    out = np.zeros(8)
    lib.gdt_smoothing_1d.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double, nb._pd]
    lib.gdt_smoothing_1d(C.byref(s), C.byref(p), -0.1, 1.1, 1, 1e-9, out.ctypes.data_as(nb._pd))
    assert out[6] == 0 and int(out[5]) & 2
    mc.raise_on_bandwidth_errors = True
    assert BandwidthError is not None


def test_smoothing_scalars_equal_the_python_expressions():
    """batch1d.hpp's smoothing_1d against MCSamples._bandwidth_1d + the smoothing block of get1DDensities on random
    parameters: every branch (solver failed, very small, fixed scales below and above 1, higher orders, periodic)."""
    from getdist_amd.batch1d import Density1DSettings
    from getdist_amd.batch2d import ParamState
    from getdist_amd.mcsamples import MCSamples

    lib = nb.harness()
    lib.gdt_smoothing_1d.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double, nb._pd]
    rng = np.random.default_rng(11)
    out = np.zeros(8)

    class Par:
        name = "p"

    class Host:  # the two methods under test need these attributes only
        raise_on_bandwidth_errors = False
        mult_bias_correction_order = 1
        no_warning_params = []
        no_warning_chi2_params = True
        _no_bandwidth_warning = MCSamples._no_bandwidth_warning

    seen = set()
    for t in range(4000):
        F = int(rng.choice([64, 256, 1024, 2048]))
        sss = float(rng.choice([-1.0, -0.5, 0.0, 0.4, 1.0, 2.5]))
        bco, mbc = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        par = Par()
        lo = rng.normal() * 10 ** rng.uniform(-2, 2)
        span = 10 ** rng.uniform(-3, 3)
        par.range_min, par.range_max = lo, lo + span
        par.param_min = lo + span * rng.uniform(-0.2, 0.3)
        par.param_max = lo + span * rng.uniform(0.7, 1.2)
        par.sigma_range = span * rng.uniform(0.02, 0.4)
        par.err = span * rng.uniform(0.02, 0.4)
        par.has_limits_bot, par.has_limits_top = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        par.periodic = bool(rng.integers(0, 8) == 0)
        N_eff = 10 ** rng.uniform(1, 7)
        have_h = bool(rng.integers(0, 6) != 0)
        h = 10 ** rng.uniform(-5, -0.5)
        fine_width, binmin, binmax = MCSamples._bin_edges(par, F)
        num_bins = int(rng.integers(20, 200))
        # the Python-planned expressions (getdist_amd/mcsamples.py: get1DDensities)
        logging.disable(logging.WARNING)
        try:
            paramrange = par.range_max - par.range_min
            width = paramrange / (num_bins - 1)
            if sss <= 0:
                bandwidth = MCSamples._bandwidth_1d(Host(), h if have_h else None, par, N_eff, mbc, bco) * (binmax - binmin)
                bandwidth = min(bandwidth, paramrange / 4)
                smooth_1D = bandwidth * abs(sss) / fine_width
            elif sss < 1.0:
                smooth_1D = sss * par.err / fine_width
            else:
                smooth_1D = sss * width / fine_width
        finally:
            logging.disable(logging.NOTSET)
        small = smooth_1D < 2
        smooth_1D = min(max(1.0, smooth_1D), F // 2)
        winw = min(int(round(2.5 * smooth_1D)), ((F - 1) if par.periodic else F) // 2 - 2)
        s = Density1DSettings(fine_bins=F, num_bins=num_bins, boundary_correction_order=bco, mult_bias_correction_order=mbc,
                              smooth_scale_1D=sss, norm=1.0, sum_w2=1.0, uncorrelated_sampler=0, raise_on_bandwidth_errors=0)
        p = ParamState(range_min=par.range_min, range_max=par.range_max, param_min=par.param_min, param_max=par.param_max,
                       sigma_range=par.sigma_range, err=par.err, neff=N_eff, has_limits_bot=par.has_limits_bot,
                       has_limits_top=par.has_limits_top, periodic=par.periodic)
        lib.gdt_smoothing_1d(C.byref(s), C.byref(p), binmin, binmax, int(have_h), h, out.ctypes.data_as(nb._pd))
        assert out[6] == 1
        assert out[1] == smooth_1D, (t, out[1], smooth_1D)
        assert int(out[3]) == winw, t
        assert bool(int(out[5]) & 4) == small, t
        if sss <= 0:
            assert out[0] == par.kde_h, t
            assert bool(int(out[5]) & 1) == (not have_h), t
        seen.add((sss <= 0, int(out[5]) & 3, small))
    assert len(seen) >= 6


def test_native_1d_route_correlated_chain_takes_the_long_route():
    """A chain whose correlation outlasts the 8-lag probe: the entry asks for that N_eff (GD_BATCH2D_NEED_NEFF), the host
    computes it by getCorrelationLength's long route, the second call succeeds; same densities as the planned route."""
    from getdist_amd.mcsamples import MCSamples

    rng = np.random.default_rng(11)
    N = 6000
    e = rng.normal(size=(N, 3))
    x = np.zeros((N, 3))
    for i in range(1, N):
        x[i] = 0.97 * x[i - 1] + e[i]
    kw = dict(samples=x, names=["a", "b", "c"])
    ref = MCSamples(_context_factory=nb.PlainContext, **kw)
    mc = MCSamples(_context_factory=nb.HarnessContext, **kw)
    same1d(mc.get1DDensities(), ref.get1DDensities())
    assert [p.N_eff_kde for p in mc.paramNames.names] == [p.N_eff_kde for p in ref.paramNames.names]


def test_native_1d_route_uncorrelated_sampler(zoo):
    """sampler = "nested": N_eff = norm^2 / sum w^2 for every parameter (chains.py:507-508), decided inside the entry."""
    fx = zoo["block10_weighted"]
    ref = make(fx, nb.PlainContext, sampler="nested")
    mc = make(fx, nb.HarnessContext, sampler="nested")
    nb.CALLS.clear()
    same1d(mc.get1DDensities([1, 2, 7]), ref.get1DDensities([1, 2, 7]))
    assert not any(c[0] in ("autocov_lags_batch", "kde_lag_sums_batch") for c in nb.CALLS)
    assert [mc.paramNames.names[j].N_eff_kde for j in (1, 2, 7)] == [ref.paramNames.names[j].N_eff_kde for j in (1, 2, 7)]


def test_native_1d_route_solver_failure_falls_back_or_raises(zoo, monkeypatch, caplog):
    """The ISJ solver returning None (mcsamples.py:1258-1268): rule-of-thumb width + the reference's two warnings, the same
    density as the Python-planned sequence; with raise_on_bandwidth_errors the entry's GD_ERR_SOLVER becomes BandwidthError
    naming the parameter."""
    import fake_ctx
    from getdist_amd.mcsamples import BandwidthError

    fx = zoo["c1_bounded"]
    real = fake_ctx.FakeContext.isj1d

    def failing(self, hist, neff):
        h, status = real(self, hist, neff)
        status[0] = -5  # the first histogram of the call: "zero f in _bandwidth_fixed_point"
        return h, status

    monkeypatch.setattr(fake_ctx.FakeContext, "isj1d", failing)
    ref = make(fx, nb.PlainContext)
    mc = make(fx, nb.HarnessContext)
    with caplog.at_level(logging.WARNING):
        native = mc.get1DDensities([2, 0])
    assert any("1D auto bandwidth failed" in r.getMessage() for r in caplog.records)
    assert any("very small or failed" in r.getMessage() for r in caplog.records)
    same1d(native, ref.get1DDensities([2, 0]))
    assert mc.paramNames.names[2].kde_h == ref.paramNames.names[2].kde_h
    strict = make(fx, nb.HarnessContext)
    strict.raise_on_bandwidth_errors = True
    with pytest.raises(BandwidthError) as e:
        strict.get1DDensities([2, 0])
    assert fx["names"][2] in str(e.value) and "column" not in str(e.value)


def test_native_1d_route_fallback_messages_and_no_warning_params(zoo, monkeypatch, caplog):
    """The messages of getAutoBandwidth1D (mcsamples.py:1259-1266) character by character on both routes: the solver's own
    width, N_eff and the fallback width in the warning and in the BandwidthError; a parameter in ``no_warning_params`` (or a
    chi2 / minuslog parameter under ``no_warning_chi2_params``) neither warns nor raises and still takes the fallback."""
    import fake_ctx
    from getdist_amd.mcsamples import BandwidthError

    fx = zoo["c1_bounded"]
    real = fake_ctx.FakeContext.isj1d

    def failing(self, hist, neff):
        h, status = real(self, hist, neff)
        status[0] = -5  # solver returned None for the first histogram
        if len(h) > 1:
            h[1] = 1e-9  # "very small" for the second
        return h, status

    monkeypatch.setattr(fake_ctx.FakeContext, "isj1d", failing)

    def messages(ctxcls, **attrs):
        mc = make(fx, ctxcls)
        for k, v in attrs.items():
            setattr(mc, k, v)
        caplog.clear()
        with caplog.at_level(logging.WARNING):
            d = mc.get1DDensities([2, 0])
        return mc, d, [r.getMessage() for r in caplog.records if "very small or failed" in r.getMessage()]

    ref, dref, mref = messages(nb.PlainContext)
    mc, dnat, mnat = messages(nb.HarnessContext)
    assert len(mref) == 2 and mnat == mref  # the same text, digits included
    p2, p0 = ref.paramNames.names[2], ref.paramNames.names[0]
    assert mref[0] == f"auto bandwidth for {p2.name} very small or failed (h=None,N_eff={p2.N_eff_kde}). Using fallback (h={p2.kde_h})"
    assert mref[1] == f"auto bandwidth for {p0.name} very small or failed (h=1e-09,N_eff={p0.N_eff_kde}). Using fallback (h={p0.kde_h})"
    same1d(dnat, dref)
    # the raise carries the same message on both routes
    texts = []
    for ctxcls in (nb.PlainContext, nb.HarnessContext):
        strict = make(fx, ctxcls)
        strict.raise_on_bandwidth_errors = True
        with pytest.raises(BandwidthError) as e:
            strict.get1DDensities([2, 0])
        texts.append(str(e.value))
    assert texts[0] == texts[1] == mref[0]
    # silenced parameters: no warning, no error (even with raise_on_bandwidth_errors), the same fallback density
    quiet = [fx["names"][2], fx["names"][0]]
    for ctxcls in (nb.PlainContext, nb.HarnessContext):
        mq, dq, msgs = messages(ctxcls, no_warning_params=quiet, raise_on_bandwidth_errors=True)
        assert msgs == []
        same1d(dq, dref)
    # one silenced, one not: the other one still raises, naming itself
    for ctxcls in (nb.PlainContext, nb.HarnessContext):
        part = make(fx, ctxcls)
        part.no_warning_params = [fx["names"][2]]
        part.raise_on_bandwidth_errors = True
        with pytest.raises(BandwidthError) as e:
            part.get1DDensities([2, 0])
        assert str(e.value) == mref[1]
