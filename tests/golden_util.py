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


def mutator_checks(factory=None, tol=1e-9):
    """
    Every mutator of the sample set the reference offers on this path (thin, weighted_thin, filter,
    reweightAddingLogLikes, cool, removeBurn, addDerived, deleteFixedParams), getLikeStats, the autocorrelation entry
    points with many lags / vectors / corr=, and the public functions of getdist_amd.convolve, against what the imported
    reference returned for the same seeded inputs (tests/golden/mutators.npz, made by make_golden.py --mutators).
    ``factory`` = a Context stand-in for the CPU tier (None: the HIP library).
    """
    import sys

    from getdist_amd import convolve as conv
    from getdist_amd.mcsamples import MCSamples

    sys.path.insert(0, GOLDEN_DIR)
    from make_golden_inputs import mutator_inputs

    g = np.load(GOLDEN_DIR + "/mutators.npz")
    samples, weights, loglikes, names, offsets, extra = mutator_inputs()
    kw = {} if factory is None else dict(_context_factory=factory)
    cut = lambda a: [a[lo:hi] for lo, hi in zip(offsets[:-1], offsets[1:])]  # noqa: E731

    def fresh():
        return MCSamples(samples=[c.copy() for c in cut(samples)], weights=[c.copy() for c in cut(weights)],
                         loglikes=[c.copy() for c in cut(loglikes)], names=names, ranges={"m3": (-3.0, None)}, **kw)

    def check_state(mc, tag):
        assert mc.numrows == int(g[tag + "/numrows"]), tag
        assert abs(mc.norm - float(g[tag + "/norm"])) <= tol * float(g[tag + "/norm"]), tag
        assert relerr(mc.means, g[tag + "/means"]) < tol, tag
        assert relerr(mc.fullcov, g[tag + "/cov"]) < tol, tag
        assert abs(mc.max_mult - float(g[tag + "/max_mult"])) <= tol * float(g[tag + "/max_mult"]), tag
        if tag + "/loglike_sum" in g.files:
            w = mc.weights if mc.weights is not None else np.ones(mc.numrows)
            assert abs(np.dot(w, mc.loglikes) - float(g[tag + "/loglike_sum"])) <= tol * abs(float(g[tag + "/loglike_sum"])), tag
        if tag + "/chain_offsets" in g.files and tag not in ("thin3", "burn"):  # the reference leaves those two stale
            assert np.array_equal(np.asarray(mc.chain_offsets), g[tag + "/chain_offsets"]), tag

    mc = fresh()
    check_state(mc, "base")
    ls = mc.getLikeStats()
    got = np.array([ls.logLike_sample, np.nan if ls.logMeanInvLike is None else ls.logMeanInvLike, ls.meanLogLike,
                    ls.logMeanLike, ls.complexity, ls.varLogLike])
    want = g["likestats/scalars"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=tol, atol=0)
    assert np.array_equal(np.array([p.bestfit_sample for p in mc.paramNames.names]), g["likestats/bestfit"])
    # the N-D region: the reference keeps an arbitrary subset of the rows tied with the threshold likelihood, the device
    # excludes them all; without ties (this fixture) the limits are the same sample values
    assert np.array_equal(np.array([p.ND_limit_bot for p in mc.paramNames.names]), g["likestats/ND_bot"])
    assert np.array_equal(np.array([p.ND_limit_top for p in mc.paramNames.names]), g["likestats/ND_top"])
    assert mc.getLikeStats() is ls  # cached until the samples change
    for j in (0, 3):
        for wu in (True, False):
            assert relerr(mc.getAutocorrelation(j, maxOff=700, weight_units=wu), g["autocorr/%d/%d" % (j, wu)]) < tol
        assert abs(mc.getCorrelationLength(j) - float(g["corrlen/%d" % j])) < tol * float(g["corrlen/%d" % j])
    vec = mc.samples[:, 1] * mc.samples[:, 2]
    assert relerr(mc.getAutocorrelation(vec, maxOff=40, normalized=False), g["autocorr/vec"]) < tol
    assert abs(mc.getCorrelationLength(0, corr=g["autocorr/3/1"]) - float(g["corrlen/given"])) < 1e-12 * float(g["corrlen/given"])
    rng = np.random.default_rng(5)
    n_ar = 60000
    e = rng.standard_normal(n_ar)
    ar = np.empty(n_ar)
    ar[0] = e[0]
    for t in range(1, n_ar):
        ar[t] = 0.9985 * ar[t - 1] + e[t]
    slow = MCSamples(samples=ar.reshape(-1, 1), names=["s"], **kw)
    assert float(g["slow/corrlen"]) > 1000  # far beyond the direct-lag window: the FFT route
    assert abs(slow.getCorrelationLength(0) - float(g["slow/corrlen"])) < tol * float(g["slow/corrlen"])
    assert abs(slow.getCorrelationLength(0, weight_units=False) - float(g["slow/corrlen_rows"])) < tol * float(g["slow/corrlen_rows"])
    mc = fresh(); mc.thin(3); check_state(mc, "thin3")  # noqa: E702
    assert mc.weights is None and mc.chain_offsets[-1] == mc.numrows
    mc = fresh(); mc.weighted_thin(2); check_state(mc, "wthin2")  # noqa: E702
    mc = fresh(); mc.filter(mc.samples[:, 0] > -0.4); check_state(mc, "filter")  # noqa: E702
    mc = fresh(); mc.reweightAddingLogLikes(extra.copy()); check_state(mc, "reweight")  # noqa: E702
    mc = fresh(); mc.cool(1.7); check_state(mc, "cool")  # noqa: E702
    assert mc.cooled == 1.7
    mc = fresh(); mc.removeBurn(0.2); check_state(mc, "burn")  # noqa: E702
    mc = fresh()
    par = mc.addDerived(mc.samples[:, 0] * mc.samples[:, 1] + 0.2 * mc.samples[:, 3], "d01", label="d_{01}", range=(None, 6.0))
    check_state(mc, "derived")
    assert par.isDerived == bool(g["derived/isDerived"]) and mc.n == len(names) + 1 and mc.index["d01"] == len(names)
    assert relerr(mc.get1DDensity("d01").P, g["derived/P1d"]) < 1e-6
    p2 = mc.get2DDensity("m0", "d01").P
    assert abs(np.sum(p2) - float(g["derived/P2d_sum"])) < 2e-3 * float(g["derived/P2d_sum"])  # TNC pair: see DESIGN.md
    try:
        mc.addDerived(mc.samples[:, 0], "d01")
        raise AssertionError("duplicate derived name accepted")
    except ValueError:
        pass
    fx = np.column_stack([samples[:, 0], np.full(len(samples), 0.25), samples[:, 1]])
    mc = MCSamples(samples=fx, weights=weights.copy(), names=["a", "fixed", "b"], **kw)
    assert [p.name for p in mc.paramNames.names] == [str(v) for v in g["fixed/names_after"]]
    assert mc.ranges.getLower("fixed") == float(g["fixed/value"])
    mc2 = MCSamples(samples=samples[:, :3].copy(), weights=weights.copy(), names=["a", "b", "c"], **kw)
    mc2.samples = np.column_stack([samples[:, 0], np.full(len(samples), 0.25), samples[:, 1]])  # edited in place, then:
    fixed, values = mc2.deleteFixedParams()
    assert fixed == [1] and values == [0.25] and mc2.n == 2 and [p.name for p in mc2.paramNames.names] == ["a", "c"]
    assert relerr(mc2.means, [np.average(samples[:, 0], weights=weights), np.average(samples[:, 1], weights=weights)]) < 1e-12
    # ---- getdist_amd.convolve against getdist.convolve
    conv.set_context(mc.ctx)
    try:
        rng = np.random.default_rng(123)
        x1, y1, ys, x2, y2 = rng.random(1024), rng.random(141), rng.random(1203), rng.random((128, 128)), rng.random((31, 31))
        xl = rng.random(1500)
        for mode in ("same", "valid", "full"):
            assert relerr(conv.convolve1D(x1, y1, mode), g["conv1d/direct/" + mode]) < 1e-13, mode
            assert relerr(conv.convolve1D(xl, ys, mode, largest_size=3000), g["conv1d/fft/" + mode]) < 1e-12, mode
            assert relerr(conv.convolve2D(x2, y2, mode, largest_size=128 + 2 * 15 + 31), g["conv2d/" + mode]) < 1e-12, mode
        assert relerr(conv.convolve1D(x1, y1, "periodic"), g["conv1d/periodic"]) < 1e-12
        for mode in ("periodic", "periodic_x", "periodic_y"):
            assert relerr(conv.convolve2D(x2[:96, :80], y2[:21, :17], mode), g["conv2d/" + mode]) < 1e-12, mode
        z = rng.standard_normal(20000)
        assert relerr(conv.autoConvolve(z, 300), g["autoconv/norm"]) < 1e-11
        assert relerr(conv.autoConvolve(z, 300, normalize=False), g["autoconv/raw"]) < 1e-11
        assert relerr(conv.autoCorrelation(z, 200), g["autocorrfn"]) < 1e-11
    finally:
        conv.set_context(None)


def prefill_plot_caches_checks(zoo, factory=None, tol=1e-9):
    """getdist_amd.plotting.prefill_plot_caches: plots.MCSampleAnalysis cache layout (plots.py:594-645), keys, contour
    counts, grids and device contour levels against the oracle (CPU tier: numpy double; -m gpu: the HIP path)."""
    from getdist_amd.mcsamples import MCSamples
    from getdist_amd.plotting import prefill_plot_caches
    from oracle import kde_oracle as ko

    class Analysis:  # the two dicts of getdist.plots.MCSampleAnalysis
        def __init__(self):
            self.densities_1D, self.densities_2D = {}, {}

    fx = zoo["c1_bounded"]
    kw = {} if factory is None else dict(_context_factory=factory)
    mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], **kw)
    an = Analysis()
    n1, n2 = prefill_plot_caches(an, "chain", mc, params=fx["names"][:3], conts=2)
    assert (n1, n2) == (3, 3)
    assert set(an.densities_1D["chain"]) == {(nm, False) for nm in fx["names"][:3]}
    a, b, c = fx["names"][:3]
    assert set(an.densities_2D["chain"]) == {(a, b, False, 2), (a, c, False, 2), (b, c, False, 2)}
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    d1 = an.densities_1D["chain"][(b, False)]
    assert np.max(np.abs(d1.P - orc.density_1d(1)["P"])) < max(tol, 1e-9)
    # (a, c) of this fixture: both unbounded -> the bandwidth passes through TNC; on the device the grid is compared
    # at the device's own bandwidth (the chaotic map is covered by test_density_2d), the layout checks are exact
    d = an.densities_2D["chain"][(a, c, False, 2)]
    assert len(d.contours) == 2 and d.P.shape == (256, 256) and d.P.max() == 1.0
    tr = {}
    o = orc.density_2d(0, 2, trace=tr)
    if relerr(d.bandwidth, (tr["hx"], tr["hy"], tr["c"])) < 1e-6:
        assert np.max(np.abs(d.P - o["P"])) < 1e-6
        assert np.allclose(d.contours, ko.contour_levels(o["P"], (0.68, 0.95)), rtol=1e-6)
    assert np.allclose(d.contours, ko.contour_levels(d.P, (0.68, 0.95)), rtol=1e-9)  # device levels of the device grid
    # a second fill replaces nothing it should not and returns the cached objects to the plotting layer
    assert prefill_plot_caches(an, "chain", mc, params=fx["names"][:2], conts=1) == (2, 1)
    assert (a, b, False, 1) in an.densities_2D["chain"] and (a, b, False, 2) in an.densities_2D["chain"]


def root_constructor_checks(tmp_path, factory=None):
    """MCSamples(root=...) / loadMCSamples: text chains on the first load, the column-major binary cache afterwards
    (same arrays bit for bit, views of one page-locked block on the HIP path), invalidation by mtime, ini settings,
    and statistics / a density from the reloaded object."""
    from getdist_amd import chainfiles
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import mcmc_chains_fixture

    kw = {} if factory is None else dict(_context_factory=factory)
    samples, weights, loglikes, names, offsets = mcmc_chains_fixture(nchains=3, N=500, n=3)
    root = str(tmp_path / "run")
    for c, (a, b) in enumerate(zip(offsets[:-1], offsets[1:])):
        np.savetxt("%s_%d.txt" % (root, c + 1), np.column_stack([weights[a:b], loglikes[a:b], samples[a:b]]), fmt="%.17g")
    (tmp_path / "run.paramnames").write_text("x\ny\nz*\n")
    (tmp_path / "my.ini").write_text("# analysis settings\nignore_rows = 0.1\nfine_bins_2D = 128\nplot_ext = pdf\ncontours = 0.5 0.9\n")
    first = MCSamples(root=root, **kw)
    assert os.path.isfile(chainfiles.cache_path(root))
    assert np.array_equal(first.samples, samples) and np.array_equal(first.weights, weights)
    again = MCSamples(root=root, **kw)
    assert np.array_equal(again.samples, samples) and np.array_equal(again.weights, weights)
    assert np.array_equal(again.loglikes, loglikes) and list(again.chain_offsets) == list(offsets)
    assert again.samples.flags.f_contiguous  # a view of the cache block: no host-side copy or transpose
    assert again.paramNames.numNonDerived() == 2 and again.name_tag == "run"
    # the reloaded object computes: pooled moments, Gelman-Rubin over the three chains, a density
    assert np.allclose(again.means, weights.dot(samples) / weights.sum(), rtol=1e-12)
    assert np.array_equal(again.means, first.means) and np.array_equal(again.fullcov, first.fullcov)
    assert again.getGelmanRubin() > 0 and again.get1DDensity("x").P.max() == 1.0
    keep_alive = again.samples  # page-locked on the HIP path: must stay readable after the object is gone
    del again
    assert np.array_equal(keep_alive, samples)
    loaded = chainfiles.read_root(root)
    assert loaded["from_cache"]
    # a newer chain file invalidates the cache
    os.utime(chainfiles.cache_path(root), (1, 1))
    assert not chainfiles.read_root(root, no_cache=True)["from_cache"]
    assert not chainfiles.read_root(root)["from_cache"] and chainfiles.read_root(root)["from_cache"]
    burnt = chainfiles.loadMCSamples(root, ini=str(tmp_path / "my.ini"), **kw)
    keep = np.concatenate([np.arange(a + int(round((b - a) * 0.1)), b) for a, b in zip(offsets[:-1], offsets[1:])])
    assert np.array_equal(burnt.samples, samples[keep]) and burnt.fine_bins_2D == 128 and list(burnt.contours) == [0.5, 0.9]
    excl = chainfiles.loadMCSamples(root, chain_exclude=[2], **kw)
    assert excl.numrows == len(weights) - (offsets[2] - offsets[1]) and len(excl.chain_offsets) == 3


MASK_CORNER_KW_PERIODIC = ({"fine_bins_2D": 64}, dict(fine_bins_2D=64, mult_bias_correction_order=0),
                           dict(fine_bins_2D=32, boundary_correction_order=0, mult_bias_correction_order=2))
MASK_CORNER_KW_LIKES = ({}, dict(mult_bias_correction_order=0))


def assert_grid_or_oracle_ensemble(d, o, tr, key, tol=1e-6, cap=5e-4, oracle_at=None):
    """
    The 2D grid gate used wherever a pair may go through TNC: ``d`` the device's Density2D,
    ``o`` / ``tr`` the oracle's result and trace.  Either the grids agree to ``tol`` of the grid maximum, or the pair
    must EARN the loose gate: it went through TNC, the oracle itself moves by more than ``tol`` under rounding-size
    perturbations of its own functionals (ko.judge_triple: +-1..12e-15, widened up to 1e-12 only if needed), the device's
    raw bandwidth triple lies inside that ensemble's range or is as good in the reference's own objective (AMISE), and
    the grids still agree to ``cap``.  With ``oracle_at`` (a callable: bandwidth triple -> the oracle's grid computed with
    that triple instead of its own, OracleSamples.density_2d(..., _bandwidths=)) the cap is replaced by the closed loop:
    the oracle's grid AT THE DEVICE'S TRIPLE equals the device's grid to ``tol`` -- everything but the triple is then
    pinned at the strict tolerance, and the triple by the ensemble.
    Returns (error, loose?).
    """
    from oracle import kde_oracle as ko

    assert d.P.shape == o["P"].shape, key
    err = float(np.max(np.abs(d.P - o["P"])))
    if err <= tol:
        return err, False
    assert "p_13" in tr and d.kopt is not None, (key, "grids differ by %.2e and the pair did not go through TNC" % err)
    psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
    verdict = ko.judge_triple(d.kopt[8:11], psi, tr["opt_N"], tr["opt_corr"])
    assert verdict["moved"] > tol, (key, "grids differ by %.2e but the oracle is stable here (moves %.2e)" % (err, verdict["moved"]))
    assert verdict["ok"], (key, "device triple outside the oracle's spread and worse in AMISE", d.kopt[8:11], verdict)
    if oracle_at is not None:
        err_at = float(np.max(np.abs(d.P - oracle_at(d.bandwidth))))
        assert err_at <= tol, (key, "oracle grid at the device's bandwidth triple", err_at)
    else:
        assert err < cap, (key, err)
    return err, True


def triangle_plot_golden_checks(zoo, factory=None):
    """tests/golden/triangle_plot_levels.npz holds what GetDist's REAL plotter (plots.triangle_plot through
    MCSampleAnalysis.get_density / get_density_grid, plots.py:594-645,2845-2878) drew from caches filled by
    getdist_amd.plotting.prefill_plot_caches, with the per-pair getters patched to raise (scripts/drive_real_caller.py, build
    container, reference imported; also compared there with the figure the reference draws of the same samples by itself).
    Here the same caches are filled on this tier's context (numpy double / HIP path) and compared with that record: the
    contour levels the plotter drew, the grids and 1D curves it read."""
    import os

    from getdist_amd.mcsamples import MCSamples
    from getdist_amd.plotting import prefill_plot_caches

    gold = np.load(os.path.join(GOLDEN_DIR, "triangle_plot_levels.npz"))
    fx = zoo["c1_bounded"]
    params = fx["names"][:4]
    kw = {} if factory is None else dict(_context_factory=factory)
    mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], **kw)

    class Analysis:
        def __init__(self):
            self.densities_1D, self.densities_2D = {}, {}

    an = Analysis()
    assert prefill_plot_caches(an, "amd_chain", mc, params=params, conts=2) == (4, 6)
    exact = factory is not None  # the record was made on the numpy double: that tier reproduces it to rounding
    for i, a in enumerate(params):
        d1 = an.densities_1D["amd_chain"][(a, False)]
        assert np.max(np.abs(d1.P[::8] - gold["cache1d/%s/P8" % a])) < (1e-12 if exact else 1e-6), a
        # the curve the plotter drew IS the cached density (plots.py:1760-1790 normalises to the maximum, already 1)
        assert np.max(np.abs(np.interp(gold["1d/%s/x" % a], d1.x, d1.P) - gold["1d/%s/y" % a])) < 1e-6, a
        for b in params[i + 1:]:
            d2 = an.densities_2D["amd_chain"][(a, b, False, 2)]
            drawn = gold["2d/%s/%s/levels" % (a, b)]
            # matplotlib drew the cache's levels (ascending, closed by an upper level above the maximum of 1)
            lv = np.sort(np.asarray(d2.contours, dtype=float))
            assert drawn.size == lv.size + 1 and drawn[-1] > 1.0
            # (a pair whose bandwidth goes through TNC follows rounding to ~1e-5: DESIGN.md section 4)
            assert np.allclose(lv, drawn[:-1], rtol=(1e-12 if exact else 1e-4)), (a, b, lv, drawn)
            assert np.max(np.abs(d2.P[::16, ::16] - gold["cache2d/%s/%s/P16" % (a, b)])) < (1e-12 if exact else 2e-5), (a, b)
            lims = gold["2d/%s/%s/lims" % (a, b)]
            assert d2.x[0] <= lims[0] + 1e-9 and d2.x[-1] >= lims[1] - 1e-9 and d2.y[0] <= lims[2] + 1e-9 and d2.y[-1] >= lims[3] - 1e-9
