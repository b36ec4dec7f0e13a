"""
CPU tests of the HOST layer (getdist_amd/mcsamples.py, bench.one_step, parallel.py) with the C ABI replaced by the
numpy test double in tests/fake_ctx.py: ranges/limits, branch selection, batching by grid and frame class, the
asynchronous TNC pool, result assembly and the 2-rank partition are exercised without a GPU.  (The kernels themselves
are covered by the -m gpu tests.)
"""

import os
import socket
import subprocess
import sys
import textwrap

import pytest
import numpy as np

import golden_util as gu
from fake_ctx import FakeContext
from oracle import kde_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(fx, **kw):
    from getdist_amd.mcsamples import MCSamples

    return MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"],
                     _context_factory=FakeContext, **kw)


def test_host_logic_matches_oracle_1d_and_2d(zoo):
    fx = zoo["c1_bounded"]
    mc = make(fx)
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    assert np.allclose(mc.fullcov, orc.fullcov, rtol=1e-12)
    for j, d in enumerate(mc.get1DDensities()):
        o = orc.density_1d(j)
        assert np.max(np.abs(d.P - o["P"])) < 1e-9
        assert np.allclose([d.x[0], d.x[-1]], [o["x"][0], o["x"][-1]], rtol=0, atol=0)
    dens = mc.get2DDensities(fx["pairs"], get_density=False)
    for (a, b), d in zip(fx["pairs"], dens):
        assert np.allclose(d.contours, ko.contour_levels(d.P, tuple(mc.contours)), rtol=1e-12)
        tr = {}
        o = orc.density_2d(a, b, trace=tr)
        assert d.bandwidth_branch == tr["branch"]
        assert np.allclose(d.bandwidth, (tr["hx"], tr["hy"], tr["c"]), rtol=1e-9)  # bounded pairs: no TNC
        assert np.max(np.abs(d.P - o["P"])) < 1e-9


def test_host_logic_meanlikes(zoo):
    """The meanlikes branches of the 1D/2D host path (weights swap, likes kernels' call order) against the goldens."""
    from getdist_amd.mcsamples import MCSamples

    g = np.load(gu.GOLDEN_DIR + "/meanlikes.npz")
    for case, kw1, kw2, fx, ll in gu.meanlikes_cases(zoo, g):
        if not case.startswith(("c1_bounded/default", "periodic")):
            continue
        mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"],
                       loglikes=ll, _context_factory=FakeContext)
        for shade in (False, True):
            mc.shade_likes_is_mean_loglikes = shade
            for j in range(min(6, len(fx["names"]))):
                d = mc.get1DDensityGridData(j, meanlikes=True, **kw1)
                assert gu.relerr(d.likes, g["%s/1d/%d/shade%d" % (case, j, shade)]) < 1e-9, (case, j, shade)
        mc.shade_likes_is_mean_loglikes = False
        for a, b in fx["pairs"][:3]:
            d = mc.get2DDensityGridData(a, b, meanlikes=True, **kw2)
            st = int(g["%s/2d/%d_%d/stride" % (case, a, b)])
            bad = gu.likes_outliers(d.likes[::st, ::st], g["%s/2d/%d_%d/likes" % (case, a, b)])
            assert bad <= gu.MAX_LIKES_OUTLIERS, (case, a, b, bad)
        assert mc.ctx._w_sel == 0  # the sample weights are selected again
        assert mc.get1DDensityGridData(0).likes is None


def test_host_logic_where_weights_and_vectors(zoo):
    """chains.py:325-337,636-838: vector arguments, `where=` filters and alternative weights, against plain numpy."""
    fx = zoo["block10_weighted"]
    mc = make(fx)
    s, w = np.asarray(fx["samples"]), np.asarray(fx["weights"])
    where = s[:, 0] > 0.5
    ww = w[where]
    assert np.isclose(mc.get_norm(where), ww.sum(), rtol=1e-13)
    assert np.isclose(mc.mean(2, where), ww.dot(s[where, 2]) / ww.sum(), rtol=1e-12)
    d = s[where, 2] - ww.dot(s[where, 2]) / ww.sum()
    assert np.isclose(mc.var(2, where), ww.dot(d * d) / ww.sum(), rtol=1e-12)
    sub = s[where][:, [1, 3, 4]]
    dm = sub - ww.dot(sub) / ww.sum()
    assert np.allclose(mc.cov([1, 3, 4], where), (dm * ww[:, None]).T @ dm / ww.sum(), rtol=1e-11)
    idx = np.nonzero(where)[0][::3]  # an index array filters like x[idx]
    assert np.isclose(mc.mean(2, idx), w[idx].dot(s[idx, 2]) / w[idx].sum(), rtol=1e-12)
    vec = s[:, 0] ** 2 + s[:, 1]
    assert np.isclose(mc.mean(vec), w.dot(vec) / w.sum(), rtol=1e-12)
    assert np.isclose(mc.std(vec), np.sqrt(w.dot((vec - w.dot(vec) / w.sum()) ** 2) / w.sum()), rtol=1e-12)
    c = mc.cov([vec, 1])
    dv = np.column_stack([vec, s[:, 1]])
    dv = dv - w.dot(dv) / w.sum()
    assert np.allclose(c, (dv * w[:, None]).T @ dv / w.sum(), rtol=1e-11)
    assert np.isclose(mc.mean(-2), w.dot(w) / w.sum(), rtol=1e-12)  # par=-2: the weights vector
    # quantiles of a vector, over a row range, with alternative weights (chains.py:793-838)
    alt = np.abs(np.sin(np.arange(len(w)))) + 0.1
    for args in (dict(), dict(start=100, end=15000), dict(weights=alt), dict(start=7, end=9000, weights=alt)):
        a, b = args.get("start", 0), args.get("end", len(w))
        wt = args.get("weights", w)[a:b]
        x = vec[a:b]
        order = x.argsort()
        cum = np.cumsum(wt[order])
        for upper in (False, True):
            f = np.array([0.025, 0.5, 0.9])
            tgt = cum[-1] * ((1 - f) if upper else f)
            want = x[order[np.minimum(np.searchsorted(cum, tgt), len(x) - 1)]]
            assert np.array_equal(mc.confidence(vec, f, upper=upper, **args), want), args
            h = mc.initParamConfidenceData(vec, **args)
            assert np.array_equal(mc.confidence(h, f, upper=upper), want), args
    assert mc.ctx._w_sel == 0 and np.isclose(mc.mean(2), w.dot(s[:, 2]) / w.sum(), rtol=1e-12)


def nd_ranges_check(zoo, factory=None):
    """range_ND_contour = k: ND confidence-region limits and the widened ranges against the reference goldens."""
    from getdist_amd.mcsamples import MCSamples, SettingError
    from oracle.fixtures import loglikes_for

    g = np.load(gu.GOLDEN_DIR + "/nd_ranges.npz")
    kw = {} if factory is None else dict(_context_factory=factory)

    def same(a, b):  # the moments differ from the reference run's in the last bits (summation order), so err does too
        return np.allclose(a, b, rtol=1e-12, atol=0)

    for nm in ("block10_weighted", "shapes", "c1_bounded"):
        fx = zoo[nm]
        ll = loglikes_for(fx["samples"])
        for k in (0, 1, 2):
            mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"],
                           loglikes=ll, settings={"range_ND_contour": k}, **kw)
            mc._init_params(list(range(mc.n)))
            pars = mc.paramNames.names
            assert same([p.range_min for p in pars], g["%s/%d/range_min" % (nm, k)]), (nm, k)
            assert same([p.range_max for p in pars], g["%s/%d/range_max" % (nm, k)]), (nm, k)
        assert np.array_equal([p.ND_limit_bot for p in pars], g["%s/ND_limit_bot" % nm])
        assert np.array_equal([p.ND_limit_top for p in pars], g["%s/ND_limit_top" % nm])
    # without loglikes the setting is inert (`self.likeStats` is None in the reference), and the index is checked
    fx = zoo["c1_bounded"]
    a = MCSamples(samples=fx["samples"], names=fx["names"], ranges=fx["ranges"], settings={"range_ND_contour": 1}, **kw)
    b = MCSamples(samples=fx["samples"], names=fx["names"], ranges=fx["ranges"], **kw)
    a._init_params([0, 1]), b._init_params([0, 1])
    assert a.paramNames.names[1].range_min == b.paramNames.names[1].range_min
    c = MCSamples(samples=fx["samples"], names=fx["names"], ranges=fx["ranges"], loglikes=loglikes_for(fx["samples"]),
                  settings={"range_ND_contour": 7}, **kw)
    try:
        c._init_params([0])
    except SettingError:
        pass
    else:
        raise AssertionError("range_ND_contour beyond the contour list must raise SettingError")


def test_host_logic_nd_ranges(zoo):
    nd_ranges_check(zoo, FakeContext)


def raftery_lewis_check(factory=None):
    """RafteryLewis / CorrSteps / thin_indices of the product against the reference goldens (integer-weight chains)."""
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import mcmc_chains_fixture

    g = np.load(gu.GOLDEN_DIR + "/raftery_lewis.npz")
    kw = {} if factory is None else dict(_context_factory=factory)
    samples, weights, loglikes, names, offsets = mcmc_chains_fixture()
    parts = list(zip(offsets[:-1], offsets[1:]))
    mc = MCSamples(samples=[samples[a:b] for a, b in parts], weights=[weights[a:b] for a, b in parts],
                   loglikes=[loglikes[a:b] for a, b in parts], names=names, **kw)
    for tc in (0.95, 0.8):
        rl = mc.getRafteryLewis(tc)
        assert np.array_equal(np.column_stack([rl["markov_thin"], rl["thin_fac"], rl["nburn"]]), g["table/%g" % tc]), tc
        assert int(mc.RL_indep_thin) == int(g["indep_thin/%g" % tc])
    for thin in (20, 3):
        mc.corr_length_thin = 0 if thin == 20 else thin
        mc.indep_thin = 0
        got_thin, corrs = mc.getCorrSteps()
        assert got_thin == thin
        assert gu.relerr(corrs, g["corrsteps/%d" % thin]) < 1e-10, thin
    text = mc.getConvergeTests(what=("RafteryLewis", "CorrSteps"))
    assert "chain  markov_thin  indep_thin    nburn" in text and "%4i%12i%12i%12i" % (0, *g["table/0.95"][0]) in text
    assert "Parameter auto-correlations as function of step separation" in text
    # thin_indices on the reference's random multiplicities, both branches
    for k in range(6):
        w = g["thin/%d/w" % k]
        one = MCSamples(samples=np.arange(len(w), dtype=float)[:, None], weights=w, names=["x"], **kw)
        assert np.array_equal(one.thin_indices(int(g["thin/%d/factor" % k])), g["thin/%d/ix" % k]), k
    # non-integer weights: the two tests are skipped like in the reference (mcsamples.py:1039)
    frac = MCSamples(samples=[samples[a:b] for a, b in parts], weights=[weights[a:b] + 0.25 for a, b in parts],
                     names=names, **kw)
    assert frac.getConvergeTests(what=("RafteryLewis", "CorrSteps")) == ""


def test_host_logic_raftery_lewis():
    raftery_lewis_check(FakeContext)


def mask_function_check(zoo, factory=None, tol=1e-9):
    """get2DDensityGridData(mask_function=...) against the oracle (pinned to the reference) incl. the returned mask."""
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import example_mask_function

    kw = {} if factory is None else dict(_context_factory=factory)
    for nm, pairs in (("c1_bounded", [(0, 3), (2, 3)]), ("shapes", [(0, 1), (6, 7)])):
        fx = zoo[nm]
        mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], **kw)
        orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
        for a, b in pairs:
            for kws in ({}, dict(mult_bias_correction_order=0), dict(boundary_correction_order=0, mult_bias_correction_order=2)):
                tr = {}
                o = orc.density_2d(a, b, mask_function=example_mask_function, trace=tr, **kws)
                # next to the cut the linear boundary correction divides by a determinant that passes through zero
                # (mcsamples.py:1950-1957): a pixel there flips between exp(3) x and 0 under 1e-16 perturbations of the
                # bandwidth or the mask moments -- in the reference too (measured: a 1e-14 change of the TNC correlation
                # moves the grid by 7.6e-4) -- and the bias-correction round spreads it over the window.  Orders 0 must
                # agree to tol; order 1 to tol whenever it is stable, else within the reference's own sensitivity.
                for bw in (None, [(tr["hx"], tr["hy"], tr["c"])]):
                    d = mc.get2DDensities([(a, b)], mask_function=example_mask_function, _bandwidths=bw, **kws)[0]
                    assert np.array_equal(d.mask, o["mask"]) and d.mask.any() and not d.mask.all()
                    err = np.abs(d.P - o["P"])
                    same_bw = np.allclose(d.bandwidth, (tr["hx"], tr["hy"], tr["c"]), rtol=1e-6)
                    if not same_bw:
                        # a pair whose TNC result the oracle itself cannot reproduce under a 1e-15 perturbation: the grid
                        # is checked against the oracle run at the device's bandwidth triple instead
                        psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
                        assert ko.get_h_is_chaotic(psi, tr["opt_N"], tr["opt_corr"])[0], (nm, a, b, kws)
                        o2 = orc.density_2d(a, b, mask_function=example_mask_function, _bandwidths=tuple(d.bandwidth), **kws)
                        err = np.abs(d.P - o2["P"])
                    if err.max() >= tol:
                        assert kws.get("boundary_correction_order", 1) == 1, (nm, a, b, kws, err.max())
                        if same_bw:  # the reference's own sensitivity at this bandwidth
                            assert err.max() < 2e-3 and np.median(err) < 1e-5, (nm, a, b, kws, err.max())
                        else:
                            # at another bandwidth a flipped pixel may even be the grid maximum everything is normalised
                            # by: allow one overall factor and the band along the cut (the tight comparison is the
                            # injected-bandwidth round of this loop)
                            ref = o2["P"]
                            sel = ref > 1e-3
                            alpha = float(np.median(d.P[sel] / ref[sel]))
                            resid = np.abs(d.P - alpha * ref)
                            assert 0.5 < alpha < 2.0 and np.median(resid) < 1e-9, (nm, a, b, kws, alpha)
                            assert np.count_nonzero(resid > 1e-6) <= 0.3 * resid.size, (nm, a, b, kws, err.max())
                    assert np.all(d.P[d.mask] == 0)
                    if bw is not None and factory is not None:
                        # same inputs through the numpy double; only the base statistics' rounding differs (one
                        # covariance pass instead of numpy's per-column dots), which the unstable order-1 pixel amplifies
                        assert err.max() < 1e-12 or kws.get("boundary_correction_order", 1) == 1
        plain = mc.get2DDensityGridData(pairs[0][0], pairs[0][1], get_density=True)
        assert plain.mask is None


def test_host_logic_mask_function(zoo):
    mask_function_check(zoo, FakeContext)


def mask_corners_check(zoo, factory=None, tol=1e-9):
    """mask_function on periodic parameters (the pair in either orientation: circular histogram side, 'valid' mask moments)
    and together with meanlikes (the mean-likelihood grid does not see the mask), against the oracle -- itself pinned to the
    reference's stored outputs for exactly these calls (test_oracle_mask_corners_golden).  Bandwidths are injected from
    the oracle so that the comparison is of the masked convolution path alone.  mcsamples.py:1874-1903, 1907-1987."""
    import golden_util as gu
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import example_mask_function, loglikes_for

    kwf = {} if factory is None else dict(_context_factory=factory)
    fx = zoo["periodic"]
    mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], **kwf)
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for a, b in fx["pairs"]:
        for kws in gu.MASK_CORNER_KW_PERIODIC:
            tr = {}
            o = orc.density_2d(a, b, mask_function=example_mask_function, trace=tr, **kws)
            for bw in (None, [(tr["hx"], tr["hy"], tr["c"])]):
                d = mc.get2DDensities([(a, b)], mask_function=example_mask_function, _bandwidths=bw, **kws)[0]
                assert np.array_equal(d.mask, o["mask"]) and d.mask.any() and not d.mask.all()
                assert np.allclose(d.bandwidth, (tr["hx"], tr["hy"], tr["c"]), rtol=1e-6), (a, b, kws)
                err = float(np.max(np.abs(d.P - o["P"])))
                if err >= tol:
                    # only where the reference itself is unstable: the linear boundary correction next to the cut divides
                    # by a determinant that passes through zero (mcsamples.py:1950-1957; see mask_function_check) and a
                    # following bias-correction round spreads the flipped pixel over its window.  Shown, not assumed: the
                    # ORACLE's grid moves by as much under bandwidth perturbations of 1e-16 .. 1e-13.
                    assert kws.get("boundary_correction_order", 1) == 1 and kws.get("mult_bias_correction_order", 1) > 0, (a, b, kws, err)
                    sens = 0.0
                    for k, eps in enumerate((1e-16, -1e-16, 1e-15, -1e-15, 1e-14, -1e-14, 1e-13, -1e-13)):
                        pert = [tr["hx"], tr["hy"], tr["c"]]
                        pert[k % 2] *= 1 + eps
                        o2 = orc.density_2d(a, b, mask_function=example_mask_function, _bandwidths=tuple(pert), **kws)
                        sens = max(sens, float(np.max(np.abs(o2["P"] - o["P"]))))
                    assert sens > 100 * tol and err < 10 * sens, (a, b, kws, bw is None, err, sens)
                assert np.all(d.P[d.mask] == 0)
    fx = zoo["c1_bounded"]
    ll = loglikes_for(fx["samples"])
    mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], loglikes=ll, **kwf)
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"], loglikes=ll)
    for a, b in ((0, 3), (2, 3)):
        for kws in gu.MASK_CORNER_KW_LIKES:  # (orders without the unstable linear-correction pixel next to the cut)
            tr = {}
            o = orc.density_2d(a, b, meanlikes=True, mask_function=example_mask_function, trace=tr, likes_exact=True, **kws)
            d = mc.get2DDensities([(a, b)], meanlikes=True, mask_function=example_mask_function,
                                  _bandwidths=[(tr["hx"], tr["hy"], tr["c"])], **kws)[0]
            assert np.array_equal(d.mask, o["mask"])
            plain = mc.get2DDensities([(a, b)], meanlikes=True, _bandwidths=[(tr["hx"], tr["hy"], tr["c"])], **kws)[0]
            assert np.max(np.abs(d.likes - plain.likes)) < 1e-12  # the mask does not reach the mean-likelihood grid
            want = o["likes_exact"] if o.get("likes_exact") is not None else o["likes"]
            assert np.max(np.abs(d.likes - want)) < max(tol, 1e-6), (a, b, kws, float(np.max(np.abs(d.likes - want))))
            err = np.abs(d.P - o["P"])
            if kws.get("boundary_correction_order", 1) == 1 and err.max() >= tol:
                assert err.max() < 2e-3 and np.median(err) < 1e-5, (a, b, kws, err.max())  # (see mask_function_check)
            else:
                assert err.max() < tol, (a, b, kws, float(err.max()))


def test_host_logic_mask_corners(zoo):
    mask_corners_check(zoo, FakeContext)


def test_prefill_plot_caches(zoo):
    """plots.MCSampleAnalysis cache layout (plots.py:594-645): keys, contour counts, one batched call per dimension."""
    gu.prefill_plot_caches_checks(zoo, FakeContext)


def test_triangle_plot_golden(zoo):
    """The record of GetDist's real plotter drawing from the prefilled caches (tests/golden/triangle_plot_levels.npz) against
    the numpy double's caches."""
    gu.triangle_plot_golden_checks(zoo, FakeContext)


def test_real_plotter_draws_from_the_prefilled_caches():
    """scripts/drive_real_caller.py where GetDist is importable (the build container: /root/reference): the REAL
    getdist.plots.triangle_plot over MCSampleAnalysis.get_density / get_density_grid draws a 4-parameter triangle of a
    getdist_amd sample set from caches filled by two batched calls -- the per-parameter / per-pair getters raise -- and the
    levels, limits and curves equal the committed record and the figure the reference draws by itself."""
    import subprocess
    import sys

    ref = os.environ.get("GETDIST_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "getdist")):
        pytest.skip("GetDist is not available on this box (the record is checked by test_triangle_plot_golden)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "drive_real_caller.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600, env=dict(os.environ, MPLBACKEND="Agg"))
    assert out.returncode == 0 and "triangle plot drawn by getdist.plots" in out.stdout, out.stdout[-3000:]


def test_planned_route_with_and_without_the_second_context(zoo, monkeypatch):
    """The comparison route (tests/planned_route.py) runs the binning and alternate convolution batches on a second context
    (own stream over the same resident samples) for batches of >= 64 pairs: same grids, same order, same bandwidths as
    everything on one context; settings changed afterwards reach the second context too, a re-upload drops it."""
    fx = zoo["block50"]
    pairs = [(i, j) for i in range(13) for j in range(i + 1, 13)] + [(20, 21), (38, 39)]
    assert len(pairs) >= 64
    monkeypatch.setenv("GETDIST_AMD_OVERLAP_NEFF", "0")
    ref = make(fx)
    ref.CONV_TWO_STREAMS_PAIRS = (1 << 30, 0)  # and every batch convolved on this context's stream
    one = ref.get2DDensities(pairs)
    assert ref._twin is None
    monkeypatch.setenv("GETDIST_AMD_OVERLAP_NEFF", "1")  # default: the binning runs on the second context
    mc = make(fx)
    for a, b in zip(mc.get2DDensities(pairs), one):
        assert np.array_equal(a.P, b.P) and np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y)
        assert a.bandwidth_branch == b.bandwidth_branch and np.allclose(a.bandwidth, b.bandwidth, rtol=1e-12)
    assert mc._twin is not None and mc._nlanes == 1
    mc.updateSettings({"fine_bins_2D": 128})
    mc.updateBaseStatistics()
    again = mc.get2DDensities(pairs)
    assert all(d.P.shape[0] in (128, 192, 384, 576, 768, 960) for d in again) and again[0].P.shape == (128, 128)
    mc.setSamples(np.asarray(fx["samples"]) * 1.0)  # re-upload drops the second context
    assert mc._twin is None


def test_host_logic_branches_and_grid_classes(zoo):
    """block50: all three bandwidth branches, four grid sizes, bounded and unbounded pairs through the batched path."""
    fx = zoo["block50"]
    g = gu.load("block50")
    mc = make(fx)
    dens = mc.get2DDensities(fx["pairs"])
    branches = set()
    for (a, b), d in zip(fx["pairs"], dens):
        key = "p2d/%s/%s/default" % (fx["names"][a], fx["names"][b])
        assert d.P.shape[0] == int(g[key + "/F"]), key
        branches.add(d.bandwidth_branch)
        px, py = mc.paramNames.names[a], mc.paramNames.names[b]
        tnc = d.bandwidth_branch in ("A", "C") and not (px.has_limits or py.has_limits)
        gu.check_grid_2d(g, key, d.P, 2e-3 if tnc else 1e-8)
    assert branches == {"A", "B", "C"}
    # one batched histogram launch per grid-size class, not one per pair
    n_hist = [c for c in mc.ctx.log if c[0] == "hist2d_prebinned"]
    assert len(n_hist) == len({c[2] for c in n_hist}) == 4
    ms = mc.getMargeStats()
    assert len(ms.parWithName("p4").limits) == 3


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    import torch.distributed as dist
    import bench
    from fake_ctx import FakeContext
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, w, names, ranges = synth.block_recipe(10, 6000, weighted=False, stream=41)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=FakeContext)
    pairs = synth.triangle_pairs(10)
    dens = bench.one_step(mc, pairs, dist, rank, world, None)
    # which pairs did this rank take, and what did it get?
    from getdist_amd import parallel
    idx, mine = parallel.partition_pairs_by_column_blocks(pairs, bench.pair_cost_classes(mc, pairs), world, rank, mc.n)
    assert len(mine) == len(dens)
    np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), idx=np.array(idx), P=np.array([d.P for d in dens]),
             neff=np.array([p.N_eff_kde for p in mc.paramNames.names]))
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_two_rank_step_equals_single_process(tmp_path):
    """bench.one_step under gloo world_size 2: every pair computed exactly once, grids equal to the 1-rank run."""
    import bench
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GETDIST_AMD_TNC_WORKERS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0 and "rank %d ok" % rank in out, out
    s, w, names, ranges = synth.block_recipe(10, 6000, weighted=False, stream=41)
    os.environ["GETDIST_AMD_TNC_WORKERS"] = "1"
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=FakeContext)
    pairs = synth.triangle_pairs(10)
    ref = bench.one_step(mc, pairs, None, 0, 1, None)
    seen = []
    tight = 0
    for rank in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        assert np.allclose(z["neff"], [p.N_eff_kde for p in mc.paramNames.names], rtol=1e-9)  # all-gathered state
        for i, P in zip(z["idx"], z["P"]):
            seen.append(int(i))
            # the base statistics are pooled from the two ranks' row shares, so means / edges agree with the one-process
            # run to rounding, not bit for bit; pairs whose bandwidth goes through TNC may amplify that
            err = float(np.max(np.abs(P - ref[int(i)].P)))
            assert err < 2e-3, (pairs[int(i)], err)
            tight += err < 1e-8
    assert sorted(seen) == list(range(len(pairs)))
    assert tight >= 0.8 * len(pairs)


def test_chain_file_loader(tmp_path, zoo):
    """GetDist text chains + .paramnames + .ranges -> MCSamples: burn-in, fixed-parameter deletion, derived flags."""
    from getdist_amd import chainfiles
    from oracle.fixtures import mcmc_chains_fixture

    samples, weights, loglikes, names, offsets = mcmc_chains_fixture(nchains=3, N=900, n=4)
    root = str(tmp_path / "run")
    for c, (a, b) in enumerate(zip(offsets[:-1], offsets[1:])):
        block = np.column_stack([weights[a:b], loglikes[a:b], samples[a:b, :2], np.full(b - a, 0.7), samples[a:b, 2:]])
        np.savetxt("%s_%d.txt" % (root, c + 1), block, fmt="%.16e")
    (tmp_path / "run.paramnames").write_text("m0  \\mu_0\nm1\nfixedpar  f\nm2*  derived_2\nm3*\n")
    (tmp_path / "run.ranges").write_text("m0  N  N\nm1  -5.5  N\nm3  N  9\nfixedpar 0.7 0.7\n")
    (tmp_path / "other_1.txt").write_text("1 0 0\n")
    assert [os.path.basename(f) for f in chainfiles.chainFiles(root)] == ["run_1.txt", "run_2.txt", "run_3.txt"]
    mc = chainfiles.loadMCSamples(root, settings={"ignore_rows": 0.25}, _context_factory=FakeContext)
    assert mc.paramNames.list() == ["m0", "m1", "m2", "m3"]  # the fixed column is gone
    assert [p.isDerived for p in mc.paramNames.names] == [False, False, True, True]
    assert mc.paramNames.numNonDerived() == 2 and mc.paramNames.names[0].label == "\\mu_0"
    keep = np.concatenate([np.arange(a + int(round((b - a) * 0.25)), b) for a, b in zip(offsets[:-1], offsets[1:])])
    assert mc.numrows == len(keep) and len(mc.chain_offsets) == 4
    assert np.allclose(mc.samples, samples[keep], rtol=1e-15) and np.allclose(mc.weights, weights[keep])
    assert np.allclose(mc.loglikes, loglikes[keep])
    assert mc.ranges.getLower("m1") == -5.5 and mc.ranges.getUpper("m3") == 9.0 and mc.ranges.getLower("m0") is None
    assert np.isclose(mc.getGelmanRubin(), mc.getGelmanRubin()) and mc.get1DDensity("m1").P.shape == (1024,)
    full = chainfiles.loadMCSamples(root, ignore_rows=10, _context_factory=FakeContext)
    assert full.numrows == len(weights) - 30


def _expected_ingestion(kw, min_weight_ratio=1e-30):
    """Independent statement of chains.py:1017-1061,1405-1443,1488-1503 for the cases of oracle.ingestion_cases()."""
    ign = kw.get("ignore_rows", (kw.get("settings") or {}).get("ignore_rows", 0))
    lines = int(ign)
    frac = 0 if lines else ign
    is_list = isinstance(kw["samples"], list)
    chains = kw["samples"] if is_list else [kw["samples"]]
    ws = kw.get("weights") if is_list else [kw.get("weights")]
    Ls = kw.get("loglikes") if is_list else [kw.get("loglikes")]
    out = []
    for i, c in enumerate(chains):
        w = None if ws is None or ws[i] is None else ws[i]
        L = Ls[i]
        c, L = c[lines:], L[lines:]
        if w is not None:
            w = w[lines:]
            keep = w > w.max() * min_weight_ratio
            c, L, w = c[keep], L[keep], w[keep]
        k = int(round(c.shape[0] * frac))
        out.append((c[k:], None if w is None else w[k:], L[k:]))
    offsets = np.cumsum([0] + [c.shape[0] for c, _, _ in out])
    samples = np.vstack([c for c, _, _ in out])[:, [0, 1, 3]]  # column 2 never moves in chain 1 -> deleted
    weights = None if out[0][1] is None else np.hstack([w for _, w, _ in out])
    return samples, weights, np.hstack([L for _, _, L in out]), (offsets if is_list else None)


def test_array_ingestion_follows_the_reference_per_chain():
    """burn-in, the minimum-weight filter and chain offsets are per chain; fixed parameters are decided on
    the first chain and recorded as zero-width ranges; the ignore_rows setting is honoured as a fraction.  The same
    cases are compared with the imported reference in oracle/validate_against_reference.py::compare_array_ingestion."""
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import ingestion_cases

    for label, kw in ingestion_cases():
        mc = MCSamples(_context_factory=FakeContext, **kw)
        samples, weights, loglikes, offsets = _expected_ingestion(kw)
        assert mc.paramNames.list() == ["a", "b", "d"], label
        assert np.array_equal(mc.samples, samples), label
        assert (weights is None and mc.weights is None) or np.array_equal(mc.weights, weights), label
        assert np.array_equal(mc.loglikes, loglikes), label
        assert mc.ranges.getLower("c") == 0.25 and mc.ranges.getUpper("c") == 0.25, label
        if offsets is None:
            assert mc.chain_offsets is None
        else:
            assert list(mc.chain_offsets) == list(offsets), label
            assert mc.chain_offsets[-1] == mc.numrows
            D = mc.getGelmanRubinEigenvalues()  # per-chain statistics read the right row ranges
            st = mc.getSeparateChainStats(3)
            for (a, b), (cm, _, _) in zip(zip(offsets[:-1], offsets[1:]), st):
                w = np.ones(b - a) if weights is None else weights[a:b]
                assert np.allclose(cm, w @ samples[a:b] / w.sum(), rtol=1e-12)
            assert D is not None and D.shape == (3,)


def test_root_constructor_binary_cache_and_ini(tmp_path):
    """MCSamples(root=...) / loadMCSamples: text chains on the first load, the column-major binary cache afterwards
    (same arrays bit for bit, returned as views of one block), cache invalidation by mtime, ini settings."""
    gu.root_constructor_checks(tmp_path, FakeContext)


def test_separate_chains_chainlist_and_make_single():
    """chains.py:1446-1527: getSeparateChains() views, chainlist= sub-lists, makeSingle() on combined samples."""
    import pytest

    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import mcmc_chains_fixture

    samples, weights, loglikes, names, offsets = mcmc_chains_fixture(nchains=4, N=800, n=4)
    parts = [(samples[a:b], weights[a:b], loglikes[a:b]) for a, b in zip(offsets[:-1], offsets[1:])]
    mc = MCSamples(samples=[p[0] for p in parts], weights=[p[1] for p in parts], loglikes=[p[2] for p in parts],
                   names=names, _context_factory=FakeContext)
    chains = mc.getSeparateChains()
    assert len(chains) == 4
    for ch, (s, w, L) in zip(chains, parts):
        assert np.array_equal(ch.samples, s) and np.array_equal(ch.weights, w) and np.array_equal(ch.loglikes, L)
        m = w @ s / w.sum()
        assert np.allclose(ch.getMeans(), m, rtol=1e-12) and np.isclose(ch.norm, w.sum())
        d = s - m
        assert np.allclose(ch.getCov(3), ((d * w[:, None]).T @ d / w.sum())[:3, :3], rtol=1e-10, atol=1e-14)
        assert np.allclose(ch.getVars(), np.diag((d * w[:, None]).T @ d / w.sum()), rtol=1e-10)
        order = np.argsort(s[:, 1])
        cum = np.cumsum(w[order])
        assert ch.confidence(1, 0.3) == s[order[np.searchsorted(cum, 0.3 * w.sum())], 1]
    full = mc.getGelmanRubinEigenvalues()
    assert np.allclose(mc.getGelmanRubinEigenvalues(chainlist=chains), full, rtol=1e-10, atol=1e-15)
    sub = mc.getGelmanRubinEigenvalues(chainlist=chains[1:])
    # the reference formula on chains 1..3 with the POOLED means of all four (chains.py:1456-1466)
    from getdist_amd.parallel import gelman_rubin_from_chain_stats
    stats = []
    for s, w, _ in parts[1:]:
        m = w @ s / w.sum()
        d = s - m
        stats.append((m, (d * w[:, None]).T @ d / w.sum(), w.sum()))
    want = gelman_rubin_from_chain_stats(stats, weights @ samples / weights.sum())
    assert np.allclose(sub, want, rtol=1e-9, atol=1e-15) and not np.allclose(sub, full)
    assert np.isclose(mc.getGelmanRubin(chainlist=chains[:2]), np.max(mc.getGelmanRubinEigenvalues(chainlist=chains[:2])))
    # row filters on a chain (chains.py:665-733 with where=): x[where], w[where] of that chain
    s1, w1, _ = parts[1]
    keep = s1[:, 0] > 0.1
    ch = chains[1]
    assert np.isclose(ch.get_norm(keep), w1[keep].sum(), rtol=1e-12)
    mu = w1[keep] @ s1[keep] / w1[keep].sum()
    assert np.allclose(ch.mean([0, 2], keep), mu[[0, 2]], rtol=1e-11)
    dk = s1[keep] - mu
    assert np.allclose(ch.cov(where=keep), (dk * w1[keep, None]).T @ dk / w1[keep].sum(), rtol=1e-10, atol=1e-14)
    assert np.isclose(ch.var(1, keep), (w1[keep] * dk[:, 1] ** 2).sum() / w1[keep].sum(), rtol=1e-10)
    idx = np.array([3, 5, 5, 11])
    assert np.isclose(ch.mean(0, idx), (w1[idx] * s1[idx, 0]).sum() / w1[idx].sum(), rtol=1e-11)
    with pytest.raises(ValueError):
        mc.makeSingle()
    single = MCSamples(samples=samples, weights=weights, names=names, _context_factory=FakeContext)
    with pytest.raises(Exception):
        single.getSeparateChains()


def test_marge_stats_host_logic_against_goldens(zoo):
    """getMargeStats: the batched limit inputs + the per-contour decision against the reference's numbers
    (tests/golden/margestats_*.npz), with the device calls played by the oracle; and the oracle's limits routine against
    the package's own spline implementation."""
    from getdist_amd.densities import Density1D

    for name in ("c1_bounded", "shapes", "block10_weighted", "wj1d", "wj2d_weighted"):
        fx = zoo[name]
        g = np.load(gu.GOLDEN_DIR + "/margestats_%s.npz" % name)
        mc = make(fx)
        ms = mc.getMargeStats()
        for nm in fx["names"]:
            par = ms.parWithName(nm)
            want = g["lims/" + nm]
            got = np.array([[lim.lower, lim.upper, lim.twotail, lim.onetail_upper, lim.onetail_lower] for lim in par.limits],
                           dtype=float)
            assert np.array_equal(got[:, 2:], want[:, 2:]), (name, nm, "limit types")
            assert np.max(np.abs(got[:, :2] - want[:, :2])) < 1e-9 * max(1e-300, float(par.err)), (name, nm, got, want)
            d = mc.get1DDensity(nm)
            host = np.array(Density1D(d.x, d.P).getLimits(np.array(mc.contours)), dtype=float)
            orc = ko.density_limits_1d(d.x, d.P, mc.contours)
            assert np.array_equal(host[:, 2:], orc[:, 2:]) and np.max(np.abs(host[:, :2] - orc[:, :2])) < 1e-11 * (d.x[-1] - d.x[0])
        # the per-parameter entry point gives the same limits as the batch
        par = mc.paramNames.names[0]
        before = [(lim.lower, lim.upper, lim.limitTag()) for lim in par.limits]
        mc._setMargeLimits(par, mc.initParamConfidenceData(0))
        assert before == [(lim.lower, lim.upper, lim.limitTag()) for lim in par.limits]


def test_row_partitioned_base_statistics_pool_to_the_single_process_values():
    """updateBaseStatistics(row_share=...): per-rank moments of a row share pooled by an all-gather equal the one-process
    statistics (weighted and unit weights, uneven shares, a rank with no rows)."""
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    for weighted in (True, False):
        s, w, names, ranges = synth.block_recipe(10, 10_007, weighted=weighted, stream=51)
        ref = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=FakeContext)
        for world in (2, 3, 8):
            mcs = [MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=FakeContext) for _ in range(world)]
            per = (ref.numrows + world - 1) // world
            shares = [m._partial_moments(min(r * per, ref.numrows), min((r + 1) * per, ref.numrows)) for r, m in enumerate(mcs)]
            for r, m in enumerate(mcs):
                m.updateBaseStatistics(row_share=(r, world), exchange=lambda mine: shares)
                assert np.allclose(m.means, ref.means, rtol=1e-12, atol=1e-14)
                assert np.allclose(m.fullcov, ref.fullcov, rtol=1e-10, atol=1e-13)
                assert np.allclose(m.vars, ref.vars, rtol=1e-10) and np.isclose(m.norm, ref.norm, rtol=1e-13)
                assert np.array_equal(m._col_min, ref._col_min) and np.array_equal(m._col_max, ref._col_max)
                assert np.isclose(m._sum_w2, ref._sum_w2, rtol=1e-13) and m.max_mult == ref.max_mult
                d = m.get1DDensity(names[3])
                assert np.max(np.abs(d.P - ref.get1DDensity(names[3]).P)) < 1e-9


def test_lazy_results_pickle_and_copy(zoo):
    """Results of the batched call carry a waiter until their grid is first read; pickling and deep-copying complete
    the grid and leave the waiter (and the device context behind it) out of the state."""
    import copy
    import pickle

    fx = zoo["c1_bounded"]
    mc = make(fx)
    d = mc.get2DDensities([(0, 1), (2, 3)])
    assert d[0].__dict__.get("_wait") is not None
    clone = pickle.loads(pickle.dumps(d[0]))
    assert d[0].__dict__.get("_wait") is None and "_wait" not in clone.__dict__
    assert np.array_equal(clone.P, d[0].P) and np.array_equal(clone.x, d[0].x)
    deep = copy.deepcopy(d[1])
    assert np.array_equal(deep.P, d[1].P) and deep.P is not d[1].P
    assert abs(float(np.ravel(clone.Prob(clone.x[10], clone.y[20]))[0]) - clone.P[20, 10]) < 1e-9  # spline rebuilt on demand


def test_lazy_results_empty_grid_raises_at_every_read_of_that_grid_only(zoo, monkeypatch):
    """'no samples in bin' belongs to the grid that is empty -- every read of it raises, its siblings of the
    same call deliver, the single-pair entry points raise at the call site like the reference, and a later batched
    call completes (and releases) the previous one."""
    import pytest

    from getdist_amd.densities import DensitiesError

    fx = zoo["c1_bounded"]
    mc = make(fx)
    real = FakeContext.density2d

    def one_empty(self, d_hist, B, F, *a, **k):
        out, st = real(self, d_hist, B, F, *a, **k)
        st[0] = -4  # the first grid of every batch comes back empty
        return out, st

    monkeypatch.setattr(FakeContext, "density2d", one_empty)
    d = mc.get2DDensities([(0, 1), (0, 2), (1, 2)])
    first_pending = mc._pending_results
    empty = [q for q in range(3) if q in [ks[0] for _, _, ks, _, _, _, _ in first_pending.inflight]]
    assert 0 < len(empty) < 3
    for q in range(3):
        if q in empty:
            with pytest.raises(DensitiesError):
                d[q].P
            with pytest.raises(DensitiesError):  # again: not only the first reader learns about it
                d[q].P
        else:
            assert d[q].P.shape == (256, 256) and d[q].P.max() == 1.0
    with pytest.raises(DensitiesError):
        mc.get2DDensity(0, 1)
    with pytest.raises(DensitiesError):
        mc.get2DDensityGridData(0, 1, get_density=True)
    monkeypatch.setattr(FakeContext, "density2d", real)
    d2 = mc.get2DDensities([(0, 1), (2, 3)])
    assert first_pending.done and mc._pending_results is not first_pending  # completed by the later call
    assert d2[0].P.max() == 1.0
    mc._drop_second_lane()  # contexts going away complete what is pending first
    assert mc._pending_results is None


def test_reference_unit_tests_host_logic():
    """getdist_test.py's testFileLoadPlot / testLimits numbers with the device calls played by the numpy double."""
    gu.reference_unit_test_checks(FakeContext)


def test_mutators_like_stats_autocorrelation_and_convolve_host_logic():
    """SURVEY.md 8b state invalidation: every mutator, getLikeStats, the long-lag autocorrelation route and the
    convolve module against the reference's outputs, with the device calls played by the numpy double."""
    gu.mutator_checks(FakeContext)


def test_truncated_chain_file_is_rejected_and_ini_contour_keys_are_honoured(tmp_path, zoo):
    """a chain file whose last row is short raises like np.loadtxt (no NaN row reaches the samples or the
    binary cache); num_contours / contourN / force_twotail / max_frac_twotailN of a reference .ini are applied."""
    import pytest

    from getdist_amd import chainfiles
    from getdist_amd.mcsamples import MCSamples, _read_ini_settings

    good = tmp_path / "ok_1.txt"
    good.write_text("1 0.5 0.1 0.2\n2 0.6 0.3 0.4\n1 0.7 0.5 0.6\n")
    assert chainfiles.loadNumpyTxt(str(good)).shape == (3, 4)
    bad = tmp_path / "bad_1.txt"
    bad.write_text("1 0.5 0.1 0.2\n2 0.6 0.3 0.4\n1 0.7 0.5\n")
    with pytest.raises(ValueError):
        chainfiles.loadNumpyTxt(str(bad))
    ini = tmp_path / "a.ini"
    ini.write_text("num_contours = 2\ncontour1 = 0.5\ncontour2 = 0.9\nforce_twotail = T\nmax_frac_twotail2 = 0.02\n"
                   "fine_bins = 512\nplot_meanlikes = F\n")
    st = _read_ini_settings(str(ini))
    assert st["contours"] == [0.5, 0.9] and st["fine_bins"] == "512" and "plot_meanlikes" not in st
    fx = zoo["c1_bounded"]
    mc = MCSamples(samples=fx["samples"], names=fx["names"], ranges=fx["ranges"], ini=str(ini), _context_factory=FakeContext)
    assert list(mc.contours) == [0.5, 0.9] and mc.force_twotail is True and mc.fine_bins == 512
    mft = mc._max_frac_twotail()
    assert mft[1] == 0.02 and abs(mft[0] - np.exp(-0.5 * 0.6744897501960817**2)) < 1e-12
    ms = mc.getMargeStats()
    assert len(ms.names[0].limits) == 2 and len(ms.names[3].limits) == 2  # one limit per contour of the .ini
    with pytest.raises(Exception):
        _read_ini_settings({"num_contours": 2, "contour1": 0.5})


def test_batched_call_leaves_no_reference_cycles_of_its_own(zoo):
    """A looping caller keeps the cyclic collector off (bench.py): everything a batched call creates must
    then be freed by reference counts.  With the collector off, two calls may leave unreachable objects only from the
    test double's scipy optimiser, none defined in getdist_amd."""
    import gc

    fx = zoo["block10_weighted"]
    mc = make(fx)
    pairs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    mc.get2DDensities(pairs)
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    old_flags = gc.get_debug()
    gc.set_debug(gc.DEBUG_SAVEALL)
    try:
        first = mc.get2DDensities(pairs)
        second = mc.get2DDensities(pairs)
        del first, second
        gc.collect()
        mine = [o for o in gc.garbage
                if (getattr(o, "__module__", None) or type(o).__module__ or "").startswith("getdist_amd")]
        assert not mine, [type(o).__name__ + ":" + getattr(o, "__qualname__", "") for o in mine][:10]
    finally:
        gc.set_debug(old_flags)
        gc.garbage.clear()
        if was:
            gc.enable()


def test_array_forms_of_the_host_decisions_equal_the_scalar_ones(zoo):
    """What sits between two kernels is evaluated on arrays (bin edges of all parameters, the plan's side-car arrays, the
    dealing of pairs by class): each must be the scalar code's result element by element."""
    from getdist_amd import parallel

    fx = zoo["block50"]
    mc = make(fx)
    mc.prepareParams()
    names = mc.paramNames.names
    js = list(range(mc.n))
    bmin, bmax = mc._bin_edge_arrays(js)
    for j in js:
        for F in (256, 384, 960):
            fw, lo, hi = mc._bin_edges(names[j], F)
            assert lo == bmin[j] and hi == bmax[j] and fw == (bmax[j] - bmin[j]) / (F - 1)
    pairs = [(i, j) for i in range(12) for j in range(i + 1, 12)]
    corr = np.asarray(mc.getCorrelationMatrix())
    ranges_xy = []
    for a, b in pairs:
        ea, eb = mc._bin_edges(names[a], 256), mc._bin_edges(names[b], 256)
        ranges_xy.append((ea[2] - ea[1], eb[2] - eb[1]))
    plan = mc._bandwidth_plan(pairs, [corr[b, a] for a, b in pairs], ranges_xy, 256)
    arr = plan.arr
    code = {"A": 0, "B": 1, "C": 2}
    for k, e in enumerate(plan):
        assert arr["branch"][k] == code[e["branch"]] and bool(arr["has_limits"][k]) == bool(e["has_limits"])
        assert arr["corr"][k] == e["corr"] and arr["rangex"][k] == e["rangex"] and arr["rangey"][k] == e["rangey"]
        assert arr["neff"][k] == e["neff"]
        if e["branch"] == "C":
            assert arr["fallback_t"][k] == e["fallback_t"]
    # dealing by class on arrays == the per-pair key version
    cls = (np.arange(len(pairs)) * 7919) % 5
    for world in (2, 3, 8):
        for rank in range(world):
            a = parallel.partition_pairs_by_class(pairs, cls, world, rank)
            b = parallel.partition_pairs(pairs, dict(zip(pairs, cls.tolist())).__getitem__, world, rank)
            assert a[0] == b[0] and a[1] == b[1]
    # index arrays and lists of names / indices give the same call
    d_list = mc.get2DDensities(pairs[:9])
    d_arr = mc.get2DDensities(np.array(pairs[:9]))
    d_names = mc.get2DDensities([(names[a].name, names[b].name) for a, b in pairs[:9]])
    for x, y, z in zip(d_list, d_arr, d_names):
        assert np.array_equal(x.P, y.P) and np.array_equal(x.P, z.P)


def test_pipelined_optimiser_and_convolution_equal_the_plain_order(zoo):
    """Large calls cut the optimiser's launch in two and convolve the first part on the second stream while the second
    part is optimised: same grids, bandwidths and records as the plain order (thresholds lowered so that a 78-pair call
    takes the pipelined route on the numpy double)."""
    fx = zoo["block50"]
    pairs = [(i, j) for i in range(13) for j in range(i + 1, 13)]
    ref = make(fx)
    ref.CONV_TWO_STREAMS_PAIRS = (1 << 30, 0)
    plain = ref.get2DDensities(pairs)
    mc = make(fx)
    mc.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc.KOPT_SPLIT_MIN = 8
    calls = []
    orig = mc.ctx.kopt2d

    def counting(*a, **k):
        calls.append(a[1])
        return orig(*a, **k)

    mc.ctx.kopt2d = counting
    piped = mc.get2DDensities(pairs)
    assert mc._twin is not None and len(calls) >= 2 and sum(calls) == sum(1 for d in plain if d.bandwidth_branch != "B")
    for a, b in zip(piped, plain):
        assert np.array_equal(a.P, b.P) and a.bandwidth == b.bandwidth and a.bandwidth_branch == b.bandwidth_branch
        assert (a.kopt is None) == (b.kopt is None) and (a.kopt is None or np.array_equal(a.kopt, b.kopt, equal_nan=True))


def test_bench_cpu_baseline_and_all_pairs_census_run_on_the_numpy_double(tmp_path):
    """bench.py's post-clock legs on CPU (the C ABI replaced by tests/fake_ctx.py): the oracle baseline with its stratified
    sample, the un-extrapolated small-N triangle, and the all-pairs full-size census that hosts with >= 64 cores run (here
    forced on, workers dealt one tile of the triangle each): every pair compared, none above the tolerance, and every loose
    pair would carry its verdict record (criterion, perturbation scale, nearest ensemble member)."""
    import json
    import subprocess

    env = dict(os.environ, GETDIST_AMD_CENSUS_MIN_CORES="2", PYTHONPATH=os.path.join(ROOT, "tests"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nsamples", "20000", "--nparams", "8", "--steps", "1", "--warmup", "1",
           "--cpu-small-n", "6000", "--cpu-workers", "2", "--context-factory", "fake_ctx:FakeContext"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    par = line["parity"]
    assert par["n_pairs_on_loose_gate"] == 0 and par["n_pairs_checked"] >= 3
    full = par["full_size_census"]
    assert full["ran"] and full["pairs_compared"] == 28 and full["pairs_above_1e_6"] == 0 and full["N"] == 20000
    assert line["cpu_baseline"]["full_triangle_small_n"]["parity_census"]["pairs_compared"] == 28


def test_comm_stage_timeout_never_undercuts_the_library_watchdog(monkeypatch):
    """parallel.comm_stage_timeout: the Python-side watchdog of a communicator set-up stage must fire AFTER the library's own
    (GDHIP_COMM_TIMEOUT_S) -- a value from the environment below it is raised, since a Python-side timeout is taken to mean
    "the call did not even return after the library gave up" and the context is then treated as lost."""
    from getdist_amd import parallel

    monkeypatch.setenv("GDHIP_COMM_TIMEOUT_S", "60")
    monkeypatch.setenv("GETDIST_AMD_COMM_STAGE_TIMEOUT_S", "5")
    assert parallel.comm_stage_timeout() == 70.0
    monkeypatch.setenv("GETDIST_AMD_COMM_STAGE_TIMEOUT_S", "500")
    assert parallel.comm_stage_timeout() == 500.0
    monkeypatch.delenv("GETDIST_AMD_COMM_STAGE_TIMEOUT_S")
    assert parallel.comm_stage_timeout() == 90.0
    monkeypatch.delenv("GDHIP_COMM_TIMEOUT_S")
    assert parallel.comm_stage_timeout() == 150.0

    class Ctx:
        abandoned = 0

        def comm_abandon(self):
            self.abandoned += 1

    c = Ctx()
    assert not parallel.context_is_stuck(c)
    parallel.mark_stuck(c)
    assert parallel.context_is_stuck(c) and c.abandoned == 1


def test_like_stats_column_id_does_not_survive_a_failure():
    """_setLikeStats hands the resident loglikes column to _setNDLimits through instance state; the id is reset whatever
    happens (a later _setNDLimits must upload the CURRENT loglikes, not read a slot a re-upload has rewritten)."""
    import fake_ctx
    from getdist_amd.mcsamples import MCSamples

    rng = np.random.default_rng(5)
    s = rng.normal(size=(4000, 3))
    mc = MCSamples(samples=s, loglikes=0.5 * np.sum(s**2, axis=1), names=["a", "b", "c"], _context_factory=fake_ctx.FakeContext)

    def boom():
        raise RuntimeError("injected")

    mc._setNDLimits = boom
    try:
        mc._setLikeStats()
        raise AssertionError("the injected failure did not surface")
    except RuntimeError:
        pass
    assert mc._loglikes_col is None
