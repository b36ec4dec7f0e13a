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
FIXTURES = ["c1_100k", "c1_bounded", "block10_weighted", "block50", "shapes", "shapes_intweights", "wj2d", "wj2d_weighted", "wj1d"]


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
    assert gu.relerr(mc.fullcov, g["cov"]) < 1e-12  # cov / means rel 1e-12
    assert gu.relerr(mc.getCorrelationMatrix(), g["corr"]) < 1e-12
    fracs = g["quantile_fracs"]
    for j in range(mc.n):
        q = mc.confidence(j, fracs)
        if fx["weights"] is None or name == "shapes_intweights":
            assert np.array_equal(q, g["quantiles"][j])
        else:
            # real weights: the cumulative weight is summed in another order than np.cumsum, so a target that lies
            # within rounding of a step may pick the neighbouring sample -- per quantile: the reference's sample, or
            # the one next to it in sort order
            col = np.sort(np.asarray(fx["samples"])[:, j])
            for got, want in zip(q, g["quantiles"][j]):
                if got != want:
                    k = int(np.searchsorted(col, want))
                    assert got in col[max(k - 1, 0):k + 2], (name, j, got, want)
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
                assert gu.relerr(got[:7], want[:7]) < 1e-12, (nm, got, want)
                assert abs(got[9] - want[9]) <= 1e-9 * want[9], (nm, "N_eff", got[9], want[9])
                # kde_h rel 1e-9.  A flat shape has no fixed point: fsolve wanders over rounding noise until it
                # gives up or "converges" on it, and the width it leaves is reproducible only to ~1e-4
                flat = nm in FLAT_1D_SHAPES
                assert abs(got[10] - want[10]) <= (1e-4 if flat else 1e-9) * want[10], (nm, "kde_h", got[10], want[10])


# 1D shapes whose ISJ fixed point has no root (uniform between hard bounds and the like): see test_density_1d
FLAT_1D_SHAPES = ()


def chaotic_pair_names():
    """tests/golden/tnc_chaotic_pairs.json: the pairs on which the oracle itself is chaotic (made and re-checked on the CPU)"""
    import json
    import os

    return set(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tnc_chaotic_pairs.json")))["pairs"])


def uses_tnc(d, mc, a, b):
    """Pairs whose bandwidth passes through TNC (kde_bandwidth.py:276-299): optimiser branch, no limits."""
    px, py = mc.paramNames.names[a], mc.paramNames.names[b]
    return d.bandwidth_branch in ("A", "C") and not (px.has_limits or py.has_limits)


# Gates of the 2D optimiser, at the level measured on MI355X (profiles/r03_parity_2d.json):
TOL_TSTAR = 1e-10   # Brent stops at the same iterate: t* differs only by the rounding of the fixed-point functional
TOL_PSI = 1e-10     # psi functionals at t* (fp64 bilinear forms, different summation order than numpy)
# The reference is *chaotic* through TNC for some pairs: a 1e-15 relative perturbation of the psi functionals (what any
# other BLAS / numpy build produces) moves (hx, hy, c) by up to ~1e-3 and the grid by 1-3e-4 of its maximum.  The device
# runs the same TNC (csrc/solvers.hpp, pinned evaluation by evaluation against scipy on the CPU), so wherever the
# reference is stable the strict 1e-6 gate applies end to end; the loose gate below is granted ONLY to pairs for which
# the oracle itself is shown to move by more than 1e-6 under such a perturbation (get_h_is_chaotic), and each use is
# recorded in the parity report.
# Round 3: the loose gate is 5e-4 of the grid maximum (nothing measured exceeds 3.2e-4), AND the device's raw bandwidth
# triple must lie inside the set the oracle itself produces for rounding-equal inputs (get_h_ensemble: +-1..12e-15
# perturbations of the functionals; within_oracle_spread) or, outside it, be as good in the reference's own objective
# (amise_within_oracle_range), AND per fixture there may not be more loose pairs than pairs
# on which the oracle is chaotic.  The report also counts the pairs on which the ORACLE's TNC, fed the DEVICE's
# functionals, leaves its own result: the part of the looseness that is nothing but psi rounding.
TOL_BW = 1e-9       # kde_h, t*, (hx, hy, c) rel 1e-9 (measured <= 5e-16 on every pair that does not go through TNC)
TOL_GRID_TNC = 5e-4
TOL_BW_TNC = 0.25  # the AMISE is nearly flat in the correlation direction; the grid tolerance is the real gate
PARITY_REPORT = {}


def _write_parity_report():
    import json
    import os

    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r06_parity_2d.json"), "w") as f:
        json.dump(PARITY_REPORT, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("name", FIXTURES + ["periodic"])
def test_density_2d(zoo, name):
    fx = zoo[name]
    if not fx["pairs"]:
        pytest.skip("1D-only fixture")
    g = gu.load(name)
    mc = make(fx)
    CHAOTIC = chaotic_pair_names()
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    report = PARITY_REPORT.setdefault(name, dict(pairs=0, tnc_pairs=0, loose=[], worst_tstar=0.0, worst_psi=0.0,
                                                 worst_grid_strict=0.0, worst_grid_loose=0.0, worst_bw_strict=0.0,
                                                 tnc_pairs_chaotic_in_the_oracle=0, oracle_on_device_psi_leaves_its_result=0,
                                                 worst_excess_over_oracle_spread=0.0))
    for kw in fx["kw2"]:
        dens = mc.get2DDensities(fx["pairs"], get_density=False, **kw)
        oracle_bw = []
        for (a, b), d in zip(fx["pairs"], dens):
            key = "p2d/%s/%s/%s" % (fx["names"][a], fx["names"][b], gu.kwkey(kw))
            tr = {}
            o = orc.density_2d(a, b, trace=tr, **kw)
            oracle_bw.append((tr.get("hx"), tr.get("hy"), tr.get("c")))
            assert d.P.shape == o["P"].shape, key
            auto = key + "/hxhyc" in g.files
            tnc = auto and uses_tnc(d, mc, a, b)
            report["pairs"] += 1
            report["tnc_pairs"] += bool(tnc)
            if auto:
                assert d.bandwidth_branch == tr["branch"], key
                if d.kopt is not None and "t_star" in tr:  # the device optimiser vs the oracle's
                    e_t = abs(d.kopt[0] - tr["t_star"]) / tr["t_star"]
                    want = np.array([tr["p_02"], tr["p_20"], tr["p_11"]] + ([tr["p_00"], tr["p_13"], tr["p_31"]] if "p_13" in tr else []))
                    e_p = float(np.max(np.abs(d.kopt[1:1 + len(want)] - want) / np.abs(want)))
                    report["worst_tstar"] = max(report["worst_tstar"], float(e_t))
                    report["worst_psi"] = max(report["worst_psi"], e_p)
                    assert e_t <= TOL_TSTAR, (key, d.kopt[0], tr["t_star"])
                    assert e_p <= TOL_PSI, (key, d.kopt, want)
            bw_err = gu.relerr(d.bandwidth, (tr["hx"], tr["hy"], tr["c"])) if auto else 0.0
            # (hx, hy, c) rel 1e-9.  A pair that goes through TNC may differ by what the ORACLE differs from
            # itself for rounding-equal inputs (4 x the spread of its +-1..12e-15 ensemble, computed below) when that is more
            bw_tol = TOL_BW
            ens = None
            if tnc and "p_13" in tr and d.kopt is not None:
                # every TNC pair: what the oracle does for rounding-equal inputs, and for the device's own functionals
                psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
                ens = ko.get_h_ensemble(psi, tr["opt_N"], tr["opt_corr"])
                moved = float(np.max(np.abs(ens - ens[0])) / np.max(np.abs(ens[0])))
                report["tnc_pairs_chaotic_in_the_oracle"] += moved > 1e-6
                on_dev = np.array(ko.get_h_from_psi(tuple(d.kopt[1:7]), tr["opt_N"], tr["opt_corr"], True), dtype=float)
                report["oracle_on_device_psi_leaves_its_result"] += bool(np.max(np.abs(on_dev - ens[0])) > 1e-6 * np.max(np.abs(ens[0])))
                # (reported only) does scipy's TNC on the DEVICE's functionals land on the device's triple?
                oracle_on_device_psi_gives_device_triple = bool(
                    np.max(np.abs(on_dev - d.kopt[8:11])) <= 1e-6 * np.max(np.abs(on_dev)))
                bw_tol = min(max(TOL_BW, 4 * moved), 1e-6)
            bw_agrees = bw_err < bw_tol
            if bw_agrees and bw_err >= TOL_BW:
                report.setdefault("tnc_pairs_between_1e_9_and_the_oracles_own_spread", []).append(
                    dict(pair=key, bandwidth_error=float(bw_err), oracle_moves_by=moved))
            if not bw_agrees:
                # the loose gate must be earned: a TNC pair whose bandwidth the ORACLE cannot reproduce under a 1e-15
                # perturbation of its own inputs, and a device result inside the oracle's own spread
                assert tnc and ens is not None, (key, d.bandwidth, (tr["hx"], tr["hy"], tr["c"]))
                # (the ensemble is widened scale by scale until it admits the triple or 1e-12 is reached: ko.judge_triple)
                verdict = ko.judge_triple(d.kopt[8:11], psi, tr["opt_N"], tr["opt_corr"])
                inside, excess, amise_ok = verdict["inside"], verdict["excess"], verdict["amise_ok"]
                amise_excess, amise_range, ens_rel = verdict["amise_excess"], verdict["amise_range"], verdict["scale"]
                report["loose"].append(dict(pair=key, bandwidth_error=float(bw_err), oracle_moves_by=moved,
                                            excess_over_oracle_spread=excess, inside_oracle_spread=bool(inside),
                                            amise_excess_over_ensemble_minimum=amise_excess, ensemble_amise_range=amise_range,
                                            oracle_on_device_psi_gives_device_triple=oracle_on_device_psi_gives_device_triple,
                                            ensemble_perturbation=ens_rel,
                                            # the record the review asks for: what admitted the triple under the FROZEN
                                            # rules (ko.FROZEN_CARVE_OUT), how far the nearest ensemble member is, and
                                            # whether the strict slack of 0.25 admits it as well
                                            admitted_by=verdict["admitted_by"], nearest_member_rel=verdict["nearest_member"],
                                            admitted_at_slack_1p0_reported_only=bool(verdict["ok_at_slack_1"])))
                report["worst_excess_over_oracle_spread"] = max(report["worst_excess_over_oracle_spread"], excess)
                assert moved > 1e-6, (key, "device and oracle bandwidths differ by %.2e but the oracle is stable (moves %.2e)"
                                      % (bw_err, moved))
                # ... and the pair must be one of the committed list (frozen: tests/test_oracle_golden.py recomputes it)
                assert "%s/%s" % (name, key) in CHAOTIC, (key, "loose, but not on tests/golden/tnc_chaotic_pairs.json")
                # the chaotic map has more outcomes than 24 perturbations sample: a triple outside their range must at
                # least be as good in the reference's own objective (the AMISE floor is flat where TNC stops)
                assert verdict["ok"], (key, "device triple outside the oracle's spread and worse in AMISE", d.kopt[8:11], verdict)
                assert gu.relerr(d.bandwidth, g[key + "/hxhyc"]) < TOL_BW_TNC, (key, d.bandwidth, g[key + "/hxhyc"])
            elif auto:
                report["worst_bw_strict"] = max(report["worst_bw_strict"], float(bw_err))
                if not tnc:
                    assert gu.relerr(d.bandwidth, g[key + "/hxhyc"]) < TOL_BW, (key, d.bandwidth, g[key + "/hxhyc"])
            tol = TOL_GRID if bw_agrees else TOL_GRID_TNC
            e_grid = float(np.max(np.abs(d.P - o["P"])))
            report["worst_grid_strict" if bw_agrees else "worst_grid_loose"] = max(
                report["worst_grid_strict" if bw_agrees else "worst_grid_loose"], e_grid)
            if bw_agrees:
                assert e_grid < tol, (key, "vs oracle", e_grid)
            else:
                # closed loop: the oracle's grid computed with the DEVICE's triple (admitted by the ensemble above) instead
                # of its own equals the device's grid at the strict tolerance -- nothing but the triple is loose.  How far
                # two admitted outcomes of the reference's chaotic map put the grids apart (e_grid, reported) is a property
                # of the reference, bounded here only as a sanity check.
                e_at = float(np.max(np.abs(d.P - orc.density_2d(a, b, _bandwidths=d.bandwidth, **kw)["P"])))
                report["worst_grid_at_device_triple"] = max(report.get("worst_grid_at_device_triple", 0.0), e_at)
                assert e_at < TOL_GRID, (key, "vs the oracle at the device's triple", e_at)
                assert e_grid < ko.RAW_DIFFERENCE_CAP == 4 * TOL_GRID_TNC, (key, "vs oracle", e_grid)
            if bw_agrees and not tnc:
                gu.check_grid_2d(g, key, d.P, tol)
                assert gu.relerr(d.contours, g[key + "/contours"]) < (10 * tol), key
            else:  # the committed reference grid of a TNC pair is itself one sample of the chaotic map
                gu.check_grid_2d(g, key, d.P, TOL_GRID_TNC if bw_agrees else 4 * TOL_GRID_TNC)
            assert np.allclose([d.x[0], d.x[-1], d.y[0], d.y[-1]], g[key + "/xy"], rtol=1e-12, atol=0), key
        # no more loose pairs than pairs on which the oracle itself is chaotic
        assert len(report["loose"]) <= report["tnc_pairs_chaotic_in_the_oracle"], (name, report)
        if oracle_bw[0][0] is not None:
            # same bandwidths in -> same grids out, for every pair, at the strict tolerance
            dens = mc.get2DDensities(fx["pairs"], _bandwidths=oracle_bw, **kw)
            for (a, b), d in zip(fx["pairs"], dens):
                key = "p2d/%s/%s/%s" % (fx["names"][a], fx["names"][b], gu.kwkey(kw))
                gu.check_grid_2d(g, key, d.P, TOL_GRID)
    _write_parity_report()


def test_get_h_on_the_device_against_scipy(zoo):
    """gd_get_h (closed forms + the two TNC minimisations, one wavefront per pair) on random psi tuples against the
    oracle's scipy version.  The CPU build of the same source is bit-identical to scipy (tests/test_native_solvers.py);
    on the device only libm differs (pow / sqrt rounding), so the results agree to rounding wherever the reference's map
    is stable, and every disagreement above 1e-6 must be a tuple on which the oracle is chaotic itself."""
    import warnings

    from getdist_amd._lib import Context
    from oracle.fixtures import random_psi_tuples

    ctx = Context(0)
    cases = list(random_psi_tuples(1500, seed=31))
    out = ctx.get_h([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], [1] * len(cases))
    closed = ctx.get_h([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], [0] * len(cases))
    tight = loose = 0
    worst_tight = 0.0
    loose_errs, oracle_moves = [], []
    for (psi, N, corr), got, got0 in zip(cases, out, closed):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = np.array(ko.get_h_from_psi(psi, N, corr, True), dtype=float)
            want0 = np.array(ko.get_h_from_psi(psi, N, corr, False), dtype=float)
        assert got[3] == 0 and got0[3] == 0
        assert np.allclose(got0[:3], want0, rtol=1e-13, atol=0)
        err = float(np.max(np.abs(got[:3] - want)) / np.max(np.abs(want)))
        if err < 1e-6:
            tight += 1
            worst_tight = max(worst_tight, err)
        else:
            loose += 1
            chaotic, moved = ko.get_h_is_chaotic(psi, N, corr)
            assert chaotic, (psi, N, corr, got[:3], want, moved)
            loose_errs.append(err)
            oracle_moves.append(moved)
    PARITY_REPORT["get_h_random_tuples"] = dict(
        n=len(cases), within_1e_6=tight, chaotic_in_the_oracle=loose, worst_relative_error_of_the_stable_ones=worst_tight,
        median_error_of_the_chaotic_ones=float(np.median(loose_errs)) if loose_errs else 0.0,
        max_error_of_the_chaotic_ones=float(np.max(loose_errs)) if loose_errs else 0.0,
        median_move_of_the_oracle_under_1e_15_perturbation=float(np.median(oracle_moves)) if oracle_moves else 0.0)
    _write_parity_report()
    # the random tuples carry 20-30 % noise on every functional, which makes them more TNC-sensitive than real pairs;
    # what matters is the assertion above: no disagreement on a tuple where the reference is reproducible
    assert tight >= 0.7 * len(cases)
    ctx.close()


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
    assert gu.relerr(mc.getCorrLengths(), g["corr_lengths"]) < 1e-9
    assert "autocorrelation lengths" in mc.getConvergeTests(what=("CorrLengths",))


def test_api_extras(zoo):
    """State invalidation (chains.py:276-323), confidence handles, converge-test report."""
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    fx = zoo["c1_bounded"]
    mc = make(fx)
    d_before = mc.get1DDensity("a").P.copy()
    cd = mc.initParamConfidenceData("a")
    assert np.array_equal(mc.confidence(cd, np.array([0.1, 0.9])), mc.confidence("a", np.array([0.1, 0.9])))
    assert np.allclose(mc.mean_diff("a"), fx["samples"][:, 0] - mc.means[0])
    mc.setSamples(fx["samples"] * 2.0)  # new samples -> device mirror re-uploaded, caches dropped
    assert "a" not in mc.density1D
    assert np.allclose(mc.means, 2.0 * fx["samples"].mean(axis=0), rtol=1e-12)
    d_after = mc.get1DDensity("a")
    assert np.allclose(d_after.P, d_before, atol=5e-3)  # a pure rescaling of one column leaves the shape unchanged
    samples, weights, names, offsets = synth.config_c4(nchains=4, N=20000, n=8)
    chains = [samples[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    ws = [weights[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    m2 = MCSamples(samples=chains, weights=ws, names=names)
    txt = m2.getConvergeTests()
    assert "var(mean)/mean(var)" in txt and abs(m2.GelmanRubin - m2.getGelmanRubin()) <= 1e-12 * m2.GelmanRubin


@pytest.mark.parametrize("name", ["shapes", "c1_bounded", "block10_weighted", "wj1d", "wj2d", "wj2d_weighted"])
def test_marge_stats_golden(zoo, name):
    """getMargeStats numbers (mcsamples.py:2353-2367, 2442-2531) against the reference's."""
    fx = zoo[name]
    g = np.load(gu.GOLDEN_DIR + "/margestats_%s.npz" % name)
    mc = make(fx)
    ms = mc.getMargeStats()
    for nm in fx["names"]:
        par = ms.parWithName(nm)
        want = g["lims/" + nm]
        got = np.array([[lim.lower, lim.upper, lim.twotail, lim.onetail_upper, lim.onetail_lower] for lim in par.limits],
                       dtype=float)
        assert np.array_equal(got[:, 2:], want[:, 2:]), (nm, "limit types")
        scale = max(1e-300, float(par.err))
        assert np.max(np.abs(got[:, :2] - want[:, :2])) < 2e-5 * scale, (nm, got, want)
        assert np.allclose([par.mean, par.err], g["meanerr/" + nm], rtol=1e-11)


def test_limits_analytic_like_reference():
    """getdist_test.py:136-142: unit Gaussian with a hard upper cut at 1; limits to 2 decimal places."""
    from getdist_amd.mcsamples import MCSamples

    r = np.random.default_rng(10)
    x = r.standard_normal(3_000_000)
    x = x[x < 1][:1_500_000]
    mc = MCSamples(samples=x[:, None], names=["x"], ranges={"x": (None, 1)})
    lims = mc.getMargeStats().parWithName("x").limits
    assert abs(lims[0].lower - (-0.78828)) < 1e-2
    assert abs(lims[0].upper - 0.7954) < 1e-2
    assert abs(lims[1].lower - (-1.730)) < 1e-2


@pytest.mark.parametrize("name", ["shapes_intweights", "c1_bounded", "block10_weighted"])
def test_split_tests_golden(zoo, name):
    """SplitTest numbers of getConvergeTests (mcsamples.py:1005-1034): quantiles over row sub-ranges on the device."""
    fx = zoo[name]
    g = np.load(gu.GOLDEN_DIR + "/splittests.npz")[name]
    mc = make(fx)
    st = mc.getSplitTests()
    assert st.shape == g.shape
    if fx["weights"] is None or name == "shapes_intweights":
        assert np.allclose(st, g, rtol=1e-12, atol=1e-15)
    else:  # real weights: a knife-edge quantile pick may move by one sample
        assert np.allclose(st, g, rtol=0, atol=2e-3)
    assert "Split tests" in mc.getConvergeTests(what=("SplitTest",))


def test_meanlikes_golden(zoo):
    """Mean-likelihood profiles / grids (gd_like_weights, gd_select_weights, gd_likes1d, gd_likes2d) against the
    reference outputs and the oracle, including shade_likes_is_mean_loglikes, periodic axes and mbc 0/1/2."""
    from getdist_amd.mcsamples import MCSamples

    g = np.load(gu.GOLDEN_DIR + "/meanlikes.npz")
    for case, kw1, kw2, fx, ll in gu.meanlikes_cases(zoo, g):
        mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], loglikes=ll)
        orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"], loglikes=ll)
        mc._use_like_weights(0)
        assert abs(mc.mean_loglike - orc.mean_loglike) < 1e-12 * abs(orc.mean_loglike)
        nj = min(6, len(fx["names"]))
        for shade in (False, True):
            mc.shade_likes_is_mean_loglikes = shade
            for j, d in enumerate(mc.get1DDensities(list(range(nj)), meanlikes=True, **kw1)):
                err = gu.relerr(d.likes, g["%s/1d/%d/shade%d" % (case, j, shade)])
                assert err < TOL_GRID, (case, j, shade, err)
        mc.shade_likes_is_mean_loglikes = False
        pairs = fx["pairs"][:3]
        dens = mc.get2DDensities(pairs, meanlikes=True, **kw2)
        oracle_bw = []
        for (a, b), d in zip(pairs, dens):
            tr = {}
            exact = d.P.shape[0] <= 384  # scipy's direct convolution is the comparator; keep it to seconds
            o = orc.density_2d(a, b, trace=tr, meanlikes=True, likes_exact=exact, **kw2)
            oracle_bw.append((tr["hx"], tr["hy"], tr["c"]))
            bw_agrees = gu.relerr(d.bandwidth, oracle_bw[-1]) < 1e-6
            if not bw_agrees:
                # only where the oracle's own TNC result moves under a 1e-15 perturbation of its inputs; the grids are
                # then checked at the strict tolerance against the oracle run with the device's bandwidth triple
                assert uses_tnc(d, mc, a, b), (case, a, b)
                psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
                assert ko.get_h_is_chaotic(psi, tr["opt_N"], tr["opt_corr"])[0], (case, a, b)
                assert np.max(np.abs(d.P - o["P"])) < TOL_GRID_TNC, (case, a, b)
                o = orc.density_2d(a, b, meanlikes=True, likes_exact=exact, _bandwidths=tuple(d.bandwidth), **kw2)
            assert np.max(np.abs(d.P - o["P"])) < TOL_GRID, (case, a, b)
            if exact:  # the same algorithm by direct summation: no noise-decided pixels on either side
                err = np.max(np.abs(d.likes - o["likes_exact"]))
                assert err < TOL_GRID, (case, a, b, err)
            bad = gu.likes_outliers(d.likes, o["likes"], TOL_GRID)
            assert bad <= gu.MAX_LIKES_OUTLIERS, (case, a, b, bad)
        # identical bandwidths in -> the reference's likes grids out (up to its own noise-decided pixels)
        dens = mc.get2DDensities(pairs, meanlikes=True, _bandwidths=oracle_bw, **kw2)
        for (a, b), d in zip(pairs, dens):
            st = int(g["%s/2d/%d_%d/stride" % (case, a, b)])
            bad = gu.likes_outliers(d.likes[::st, ::st], g["%s/2d/%d_%d/likes" % (case, a, b)], TOL_GRID)
            assert bad <= gu.MAX_LIKES_OUTLIERS, (case, a, b, bad)
        # the sample weights are back: plain densities unchanged by the excursion
        d0 = mc.get1DDensityGridData(0)
        assert d0.likes is None and np.max(np.abs(d0.P - orc.density_1d(0)["P"])) < TOL_GRID


def test_reference_unit_tests_on_the_device():
    """The reference's own unit tests for this path on its own inputs, through the HIP library: GelmanRubin to 4 places,
    the cut-correlated limits to 3, and the reference's actual outputs to 1e-9 / 1e-6."""
    gu.reference_unit_test_checks()
