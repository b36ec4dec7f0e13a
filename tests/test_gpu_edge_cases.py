"""GPU edge cases: tiny / ragged sizes, single column, filtered weights, degenerate ranges, odd grid sizes."""

import numpy as np
import pytest

import golden_util as gu
from oracle import kde_oracle as ko

pytestmark = pytest.mark.gpu


def mcs(*a, **k):
    from getdist_amd.mcsamples import MCSamples

    return MCSamples(*a, **k)


@pytest.mark.parametrize("N", [37, 1000, 4097])
def test_small_and_ragged_row_counts(N):
    r = np.random.default_rng(N)
    s = r.standard_normal((N, 3)) * [1.0, 2.0, 0.3] + [0.0, 5.0, -1.0]
    w = r.integers(1, 5, N).astype(float)
    mc = mcs(samples=s, weights=w)
    orc = ko.OracleSamples(s, w)
    assert np.allclose(mc.means, orc.means, rtol=1e-12)
    assert np.allclose(mc.fullcov, orc.fullcov, rtol=1e-11, atol=1e-14)
    fr = np.array([0.05, 0.5, 0.95])
    for j in range(3):
        assert np.array_equal(mc.confidence(j, fr), orc.confidence(orc.confidence_data(s[:, j]), fr))
    if N >= 1000:
        d = mc.get1DDensities()
        for j in range(3):
            assert np.max(np.abs(d[j].P - orc.density_1d(j)["P"])) < 1e-6
        d2 = mc.get2DDensity(0, 1)
        tr = {}
        o2 = orc.density_2d(0, 1, trace=tr)
        # unbounded pair: the TNC path -- 1e-6, or the oracle-ensemble criterion where the reference's own map is chaotic
        gu.assert_grid_or_oracle_ensemble(d2, o2, tr, "N=%d" % N, oracle_at=lambda bw: orc.density_2d(0, 1, _bandwidths=bw)["P"])


def test_single_column_and_unknown_names():
    r = np.random.default_rng(1)
    x = r.standard_normal(20000)
    mc = mcs(samples=x, names=["x"])
    assert mc.n == 1 and mc.fullcov.shape == (1, 1)
    assert abs(mc.fullcov[0, 0] - x.var()) < 1e-12 * x.var()
    d = mc.get1DDensity("x")
    o = ko.OracleSamples(x[:, None], names=["x"]).density_1d(0)
    assert np.max(np.abs(d.P - o["P"])) < 1e-6
    assert mc.get1DDensity("nope") is None
    assert mc.get2DDensity("x", "nope") is None


def test_min_weight_ratio_filter_and_zero_weights():
    r = np.random.default_rng(2)
    s = r.standard_normal((5000, 2))
    w = r.random(5000)
    w[::7] = 0.0  # removed by setMinWeightRatio (chains.py:1017-1027)
    mc = mcs(samples=s, weights=w)
    keep = w > 0
    assert mc.numrows == keep.sum()
    assert np.allclose(mc.means, w[keep].dot(s[keep]) / w[keep].sum(), rtol=1e-12)


def test_fixed_parameter_is_removed_like_the_reference():
    """chains.py:1029-1045,1548-1559: a column that never moves is deleted at construction and recorded as a fixed
    (zero-width) range; asking for its density gives None like any unknown name."""
    s = np.column_stack([np.full(1000, 3.0), np.random.default_rng(0).standard_normal(1000)])
    mc = mcs(samples=s, names=["c", "x"])
    assert mc.paramNames.list() == ["x"] and mc.n == 1
    assert mc.ranges.fixedValue("c") == 3.0
    assert mc.get1DDensity("c") is None
    assert mc.get1DDensity("x").P.shape == (1024,)


@pytest.mark.parametrize("kw", [dict(fine_bins=500), dict(fine_bins=257, smooth_scale_1D=0.5), dict(num_bins=50, smooth_scale_1D=1.5)])
def test_odd_1d_grid_sizes(kw):
    r = np.random.default_rng(3)
    s = np.abs(r.standard_normal((30000, 1)))
    mc = mcs(samples=s, names=["x"], ranges={"x": (0, None)})
    orc = ko.OracleSamples(s, names=["x"], ranges={"x": (0, None)})
    d = mc.get1DDensity("x", **kw)
    o = orc.density_1d(0, **kw)
    assert d.P.shape == o["P"].shape
    assert np.max(np.abs(d.P - o["P"])) < 1e-6


@pytest.mark.parametrize("kw", [dict(fine_bins_2D=100), dict(fine_bins_2D=33, smooth_scale_2D=0.4), dict(smooth_scale_2D=2.0)])
def test_odd_2d_grid_sizes(kw):
    r = np.random.default_rng(4)
    s = np.column_stack([np.abs(r.standard_normal(30000)), r.standard_normal(30000)])
    mc = mcs(samples=s, names=["x", "y"], ranges={"x": (0, None)})
    orc = ko.OracleSamples(s, names=["x", "y"], ranges={"x": (0, None)})
    d = mc.get2DDensity("x", "y", **kw)
    o = orc.density_2d(0, 1, **kw)
    assert d.P.shape == o["P"].shape
    assert np.max(np.abs(d.P - o["P"])) < 1e-6


def test_settings_and_errors():
    from getdist_amd.mcsamples import SettingError

    r = np.random.default_rng(5)
    s = r.standard_normal((20000, 2))
    mc = mcs(samples=s, settings={"fine_bins_2D": 128, "mult_bias_correction_order": 0})
    assert mc.get2DDensity(0, 1).P.shape == (128, 128)
    with pytest.raises(SettingError):
        mc.updateSettings({"no_such_setting": 1})
    with pytest.raises(SettingError):
        mc.get2DDensity(0, 1, boundary_correction_order=2)
    with pytest.raises(SettingError):
        mc.get1DDensity(0, boundary_correction_order=3)
    mc.updateSettings({"smooth_scale_2D": 0.5})
    assert mc.get2DDensity(0, 1).P.max() == 1.0


def test_upscaled_grid_through_the_device_optimiser():
    """Both parameters bounded and strongly anti-correlated: branch C of getAutoBandwidth2D on an upscaled F=576 grid
    (mcsamples.py:1812-1819, 1396-1409) -- the only way the 2D optimiser sees F > 256."""
    r = np.random.default_rng(11)
    n = 60000
    a = r.standard_normal(4 * n)
    b = -0.95 * a + np.sqrt(1 - 0.95**2) * r.standard_normal(4 * n)
    keep = (a > -1.0) & (b < 1.2)
    s = np.column_stack([a[keep][:n], b[keep][:n]])
    rng = {"x": (-1.0, None), "y": (None, 1.2)}
    mc = mcs(samples=s, names=["x", "y"], ranges=rng)
    orc = ko.OracleSamples(s, names=["x", "y"], ranges=rng)
    tr = {}
    o = orc.density_2d(0, 1, trace=tr)
    d = mc.get2DDensities([(0, 1)])[0]
    assert tr["branch"] == "C" and o["P"].shape[0] > 256 and d.P.shape == o["P"].shape
    assert d.bandwidth_branch == "C"
    assert abs(d.kopt[0] - tr["t_star"]) <= 1e-7 * tr["t_star"]
    assert np.max(np.abs(d.P - o["P"])) < 1e-6


@pytest.mark.parametrize("rho,weighted", [(0.9, False), (0.97, True), (0.5, False)])
def test_correlated_chain_neff_matches_oracle(rho, weighted):
    """AR(1) chains: the autocorrelation scan must run past the 8-lag probe (and past one 32-lag chunk for rho=0.97) and
    the adaptive Gaussian-kernel lag scan of chains.py:541-572 must take the same path as the reference."""
    r = np.random.default_rng(int(rho * 100))
    N = 200_000
    e = r.standard_normal(N)
    x = np.empty(N)
    x[0] = e[0]
    for i in range(1, N):
        x[i] = rho * x[i - 1] + np.sqrt(1 - rho * rho) * e[i]
    y = r.standard_normal(N)
    s = np.column_stack([x, y])
    w = r.integers(1, 4, N).astype(float) if weighted else None
    mc = mcs(samples=s, weights=w, names=["x", "y"])
    orc = ko.OracleSamples(s, w, names=["x", "y"])
    for j in range(2):
        cl = mc.getCorrelationLength(j, weight_units=False)
        cl_o = orc.correlation_length(s[:, j], weight_units=False)
        assert abs(cl - cl_o) <= 1e-9 * abs(cl_o), (j, cl, cl_o)
    mc.prepareParams()
    for j in range(2):
        orc.init_param(j)
        n_o = orc.neff_1d(j)
        n_g = mc.paramNames.names[j].N_eff_kde
        assert abs(n_g - n_o) <= 1e-8 * n_o, (j, n_g, n_o)
    d = mc.get1DDensity("x")
    o = orc.density_1d(0)
    assert np.max(np.abs(d.P - o["P"])) < 1e-6


def test_use_effective_samples_2d_setting():
    """getEffectiveSamplesGaussianKDE_2d (chains.py:576-635) behind use_effective_samples_2D, iid and AR(1) inputs."""
    r = np.random.default_rng(21)
    N = 100_000
    e = r.standard_normal((N, 2))
    x = np.empty((N, 2))
    x[0] = e[0]
    for i in range(1, N):
        x[i] = 0.8 * x[i - 1] + 0.6 * e[i]
    x[:, 1] = 0.5 * x[:, 0] + x[:, 1]
    iid = np.column_stack([np.abs(r.standard_normal(N)), r.standard_normal(N)])
    for s, rng in ((x, None), (iid, {"param1": (0, None)})):
        mc = mcs(samples=s, ranges=rng, settings={"use_effective_samples_2D": True})
        orc = ko.OracleSamples(s, ranges=rng, settings={"use_effective_samples_2D": True})
        n_g = mc.getEffectiveSamplesGaussianKDE_2d(0, 1)
        n_o = orc.neff_gaussian_kde_2d(0, 1)
        assert abs(n_g - n_o) <= 1e-8 * n_o, (n_g, n_o)
    d = mc.get2DDensity(0, 1)  # bounded pair: no TNC, strict tolerance
    o = orc.density_2d(0, 1)
    assert np.max(np.abs(d.P - o["P"])) < 1e-6


def test_where_filters_alternative_weights_and_vector_arguments(zoo):
    """chains.py:325-337,636-838 through gd_set_extra_column / gd_aux_weights / gd_select_weights, against numpy."""
    fx = zoo["block10_weighted"]
    mc = mcs(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
    s, w = np.asarray(fx["samples"]), np.asarray(fx["weights"])
    where = s[:, 0] > 0.5
    ww = w[where]
    assert np.isclose(mc.get_norm(where), ww.sum(), rtol=1e-12)
    assert np.isclose(mc.mean(2, where), ww.dot(s[where, 2]) / ww.sum(), rtol=1e-11)
    sub = s[where][:, [1, 3, 4]]
    dm = sub - ww.dot(sub) / ww.sum()
    assert np.allclose(mc.cov([1, 3, 4], where), (dm * ww[:, None]).T @ dm / ww.sum(), rtol=1e-10)
    assert np.isclose(mc.var(3, where), (ww.dot(dm[:, 1] ** 2) / ww.sum()), rtol=1e-10)
    vec = s[:, 0] ** 2 + s[:, 1]
    dv = np.column_stack([vec, s[:, 1]])
    dv = dv - w.dot(dv) / w.sum()
    assert np.isclose(mc.mean(vec), w.dot(vec) / w.sum(), rtol=1e-11)
    assert np.allclose(mc.cov([vec, 1]), (dv * w[:, None]).T @ dv / w.sum(), rtol=1e-10)
    alt = np.abs(np.sin(np.arange(len(w)))) + 0.1
    f = np.array([0.025, 0.5, 0.9])
    for args in (dict(), dict(start=100, end=15000), dict(weights=alt), dict(start=7, end=9000, weights=alt)):
        a, b = args.get("start", 0), args.get("end", len(w))
        wt = args.get("weights", w)[a:b]
        x = vec[a:b]
        order = x.argsort()
        cum = np.cumsum(wt[order])
        for upper in (False, True):
            tgt = cum[-1] * ((1 - f) if upper else f)
            want_pos = np.minimum(np.searchsorted(cum, tgt), len(x) - 1)
            got = mc.confidence(mc.initParamConfidenceData(vec, **args), f, upper=upper)
            got_pos = np.searchsorted(x[order], got)
            assert np.all(np.abs(got_pos - want_pos) <= 1), (args, upper)  # summation order: knife-edge picks
            assert np.all(np.isin(got, x)), args
    # the sample weights are selected again afterwards
    assert np.isclose(mc.mean(2), w.dot(s[:, 2]) / w.sum(), rtol=1e-12)
    cov_again = mc.cov([1, 3])
    assert np.allclose(cov_again, mc.getCov(pars=[1, 3]), rtol=1e-12)


def test_range_nd_contour_widening(zoo):
    """gd_set_extra_column (loglikes) + gd_quantiles + gd_col_minmax behind range_ND_contour, vs the reference goldens."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_logic_cpu import nd_ranges_check

    nd_ranges_check(zoo)


def test_raftery_lewis_corr_steps_and_thinning():
    """gd_thin_rows / gd_binary_transitions / gd_thinned_lag_sums behind getRafteryLewis, getCorrSteps, thin_indices."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_logic_cpu import raftery_lewis_check

    raftery_lewis_check()


def test_mask_function(zoo):
    """gd_density2d_masked (direct moment sums over a user-edited prior mask) through get2DDensityGridData."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_logic_cpu import mask_function_check

    mask_function_check(zoo, tol=1e-6)


def test_mask_function_on_periodic_axes_and_with_meanlikes(zoo):
    """gd_density2d_masked through the periodic route (explicit masks summed directly, histogram side circular) and the
    masked pair's mean-likelihood grid (gd_likes2d), against the oracle.  mcsamples.py:1874-1903, 1907-1987."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_logic_cpu import mask_corners_check

    mask_corners_check(zoo, tol=1e-6)


def test_lazy_result_delivery_and_overlapping_calls(monkeypatch):
    """A large batched 2D call returns once its work is enqueued: the grids complete at their first read (a mark on
    the copy stream), they stay valid when another call is issued before anything was read, and they equal the grids of
    the eager (wait-inside-the-call) path bit for bit.  Unknown-empty grids surface their status at that first read."""
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = synth.block_recipe(13, 150_000, weighted=False, stream=23)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    pairs = synth.triangle_pairs(13)  # 78 pairs: the overlapped path needs >= 64
    first = mc.get2DDensities(pairs)
    pending = mc._pending_results
    assert pending is not None and not pending.done, "the call should have returned before its copies were waited for"
    second = mc.get2DDensities(pairs)  # issued before any grid of `first` was read
    assert mc._pending_results is not pending
    P2 = [d.P.copy() for d in second]
    P1 = [d.P.copy() for d in first]
    assert pending.done
    # the eager form is the Python-planned route's (the native entry always delivers lazily): waits inside the call
    monkeypatch.setenv("GETDIST_AMD_LAZY_RESULTS", "0")
    monkeypatch.setenv("GETDIST_AMD_NATIVE_BATCH", "0")
    eager = mc.get2DDensities(pairs)
    monkeypatch.setenv("GETDIST_AMD_NATIVE_BATCH", "1")
    assert all(d.__dict__.get("_wait") is None for d in eager)
    for a, b, c in zip(P1, P2, eager):
        assert a.max() == 1.0 and np.array_equal(a, b) and np.array_equal(a, c.P)
    # the per-pair methods are views over the batched path and read their grid before returning
    one = mc.get2DDensity(names[0], names[1])
    assert np.allclose(one.P, P1[pairs.index((0, 1))], rtol=0, atol=1e-12)  # (another FFT frame size: rounding only)
