"""
GPU tests of gd_density2d_batch, the one native entry for a batch of parameter pairs (include/gdhip.h; replaces the span
mcsamples.py:1748-2010 + getAutoBandwidth2D :1285-1419 of the reference for P pairs at once): its grids, bandwidths,
branches and optimiser records must be `array_equal` to those of the Python-planned route over the same kernels
(GETDIST_AMD_NATIVE_BATCH=0), for every setting the route serves, at fixture size and on the C3 shape; calls from two
host threads on two objects sharing a device must not disturb one another.
"""

import threading

import numpy as np
import pytest

from getdist_amd import synth

pytestmark = pytest.mark.gpu


def both_routes(monkeypatch, build, call):
    """(native results, Python-planned results) of ``call(mc)`` on two fresh objects."""
    monkeypatch.setenv("GETDIST_AMD_NATIVE_BATCH", "0")
    plain = call(build())
    monkeypatch.setenv("GETDIST_AMD_NATIVE_BATCH", "1")
    mc = build()
    native = call(mc)
    assert mc._pending_results is None or type(mc._pending_results).__name__ == "PendingBatch"
    return native, plain, mc


def same(native, plain):
    assert len(native) == len(plain)
    for k, (a, b) in enumerate(zip(native, plain)):
        assert a.P.shape == b.P.shape, k
        assert np.array_equal(a.P, b.P), (k, float(np.max(np.abs(a.P - b.P))))
        assert np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y) and a.spacing == b.spacing, k
        assert a.bandwidth_branch == b.bandwidth_branch and a.bandwidth == b.bandwidth, k
        assert (a.kopt is None) == (b.kopt is None) and (a.kopt is None or np.array_equal(a.kopt, b.kopt, equal_nan=True)), k


def mc_of(recipe):
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = recipe
    return MCSamples(samples=s, weights=w, names=names, ranges=ranges)


@pytest.mark.parametrize("kw", [dict(), dict(mult_bias_correction_order=0, boundary_correction_order=0),
                                dict(boundary_correction_order=-1, mult_bias_correction_order=2), dict(smooth_scale_2D=0.4),
                                dict(smooth_scale_2D=2.0), dict(fine_bins_2D=128)])
def test_native_entry_equals_python_planned_route(monkeypatch, kw):
    recipe = synth.block_recipe(30, 400_000, weighted=False, stream=61)
    pairs = synth.triangle_pairs(30)
    native, plain, mc = both_routes(monkeypatch, lambda: mc_of(recipe), lambda m: m.get2DDensities(pairs, **kw))
    same(native, plain)
    if not kw:
        assert {d.bandwidth_branch for d in native} == {"A", "B", "C"} and len({d.P.shape[0] for d in native}) >= 3
        # a second call on the same object (index columns and N_eff cached) and after an invalidation: same bits
        same(mc.get2DDensities(pairs), plain)
        mc.ctx.batch2d_invalidate()
        same(mc.get2DDensities(pairs[:70]), plain[:70])


def test_native_entry_weighted_and_small_calls(monkeypatch):
    recipe = synth.block_recipe(12, 300_000, weighted=True, stream=62)
    pairs = synth.triangle_pairs(12)
    # real weights: the fp64 LDS atomics of the weighted binning kernels add in an order that differs from run to run, so
    # two runs of ONE route agree to rounding only (and a chaotic TNC pair may amplify that)
    for sel in (pairs, [(3, 1), (5, 9)]):
        native, plain, _ = both_routes(monkeypatch, lambda: mc_of(recipe), lambda m: m.get2DDensities(sel))
        errs = np.array([float(np.max(np.abs(a.P - b.P))) for a, b in zip(native, plain)])
        assert [a.bandwidth_branch for a in native] == [b.bandwidth_branch for b in plain]
        assert [a.P.shape for a in native] == [b.P.shape for b in plain]
        assert np.all(errs < 5e-4) and np.sum(errs >= 1e-9) <= max(1, 0.4 * len(errs)), errs
    rec_int = list(recipe)
    rec_int[1] = np.floor(recipe[1] * 3) + 1.0  # integer multiplicities: the byte-weight kernels
    native, plain, _ = both_routes(monkeypatch, lambda: mc_of(rec_int), lambda m: m.get2DDensities(pairs))
    same(native, plain)


def test_native_entry_serves_the_optional_branches(monkeypatch):
    """Round 6: injected bandwidths, the 2D effective sample numbers, mean likelihoods and a mask callback go through
    gd_density2d_batch (settings->bandwidths / pair_neff / bandwidths_only, gd_likes2d over the call's per-pair table): same
    grids, `likes`, masks and contour levels as the Python-planned comparison route, on the device."""
    from getdist_amd.mcsamples import MCSamples

    recipe = synth.block_recipe(10, 200_000, weighted=False, stream=64)
    s, w, names, ranges = recipe
    loglikes = 0.5 * np.sum(np.asarray(s)[:, :4] ** 2, axis=1)
    pairs = synth.triangle_pairs(10)[:20]

    def build():
        return MCSamples(samples=s, weights=w, loglikes=loglikes, names=names, ranges=ranges)

    def extras_same(native, plain):
        for k, (a, b) in enumerate(zip(native, plain)):
            assert (a.likes is None) == (b.likes is None), k
            # (the like-weighted histograms are sums of real weights by fp64 LDS atomics: two runs agree to rounding)
            assert a.likes is None or np.allclose(a.likes, b.likes, rtol=1e-11, atol=1e-13), (k, float(np.max(np.abs(a.likes - b.likes))))
            assert (a.mask is None) == (b.mask is None) and (a.mask is None or np.array_equal(a.mask, b.mask)), k

    auto = build().get2DDensities(pairs)
    triples = [(1.1 * d.bandwidth[0], 0.9 * d.bandwidth[1], 0.8 * d.bandwidth[2]) for d in auto]
    native, plain, _ = both_routes(monkeypatch, build, lambda m: m.get2DDensities(pairs, _bandwidths=triples))
    same(native, plain)
    assert [d.bandwidth for d in native] == [tuple(t) for t in triples]

    def with_2d_neff(m):
        m.use_effective_samples_2D = True
        return m.get2DDensities(pairs)

    native, plain, _ = both_routes(monkeypatch, build, with_2d_neff)
    same(native, plain)

    native, plain, _ = both_routes(monkeypatch, build, lambda m: m.get2DDensities(pairs, meanlikes=True))
    same(native, plain)
    extras_same(native, plain)
    assert all(d.likes is not None for d in native)

    def mask_function(minx, miny, stepx, stepy, mask):
        ny, nx = mask.shape
        yy, xx = np.arange(ny)[:, None], np.arange(nx)[None, :]
        mask[(yy - ny // 2) > 1.5 * (xx - nx // 2) + 10] = 0

    for kw in (dict(), dict(meanlikes=True), dict(get_density=False, num_plot_contours=2)):
        native, plain, _ = both_routes(monkeypatch, build, lambda m: m.get2DDensities(pairs[:6], mask_function=mask_function, **kw))
        same(native, plain)
        extras_same(native, plain)
        assert all(d.mask is not None and d.mask.any() for d in native)
        if "num_plot_contours" in kw:
            for a, b in zip(native, plain):
                assert np.allclose(a.contours, b.contours, rtol=1e-12, atol=0)


def test_native_entry_counter_wrap_in_the_second_binning_launch(monkeypatch):
    """A large call bins its main class in two launches; a 16-bit counter that wraps in the second one (the triangle's last
    pair: two columns with 70 % of their samples on one value) redoes the class with wider counters while the first
    optimiser part is already running on the first buffer's rows.  Same grids as the comparison route."""
    from getdist_amd.mcsamples import MCSamples

    rng = np.random.default_rng(21)
    N, n = 200_000, 30
    x = rng.standard_normal((N, n))
    spike = rng.random(N) < 0.7
    x[:, n - 2] = np.where(spike, -0.5, x[:, n - 2])
    x[:, n - 1] = np.where(spike, 0.25, x[:, n - 1])
    names = ["q%d" % i for i in range(n)]
    pairs = synth.triangle_pairs(n)
    assert len(pairs) > 400 and tuple(pairs[-1]) == (n - 2, n - 1)  # (a staged call: above CONV_TWO_STREAMS_PAIRS[1])
    native, plain, _ = both_routes(monkeypatch, lambda: MCSamples(samples=x, names=names), lambda m: m.get2DDensities(pairs))
    same(native, plain)
    assert float(native[-1].P.max()) == 1.0


def test_native_entry_contour_levels(monkeypatch):
    recipe = synth.block_recipe(10, 200_000, weighted=False, stream=63)
    pairs = synth.triangle_pairs(10)
    native, plain, _ = both_routes(monkeypatch, lambda: mc_of(recipe),
                                   lambda m: m.get2DDensities(pairs, get_density=False, num_plot_contours=2))
    same(native, plain)
    for a, b in zip(native, plain):  # (the level kernel accumulates masses with fp64 atomics: equal to rounding)
        assert np.allclose(a.contours, b.contours, rtol=1e-12, atol=0)


def test_native_entry_c3_shape(monkeypatch):
    """The bench's shape: 50 parameters, 1225 pairs (N = 2e6 here; the full row count runs in test_gpu_fullsize.py and in
    bench.py): the pipelined two-stream route with the optimiser's launch cut in two."""
    recipe = synth.config_c3(2_000_000, 50)
    pairs = synth.triangle_pairs(50)
    native, plain, mc = both_routes(monkeypatch, lambda: mc_of(recipe), lambda m: m.get2DDensities(pairs))
    same(native, plain)
    census = {}
    for d in native:
        census[(d.bandwidth_branch, d.P.shape[0])] = census.get((d.bandwidth_branch, d.P.shape[0]), 0) + 1
    assert len(census) >= 6, census


def test_native_entry_c3_shape_with_a_slow_second_half_of_the_deferred_chain(monkeypatch):
    """The last base part waits for the chain's SHEARED HISTOGRAMS only (round 6); the up-scaled grid classes the chain bins
    behind them may still be running when that part's bandwidths are final -- the last report, which convolves the
    rule-of-thumb pairs of those classes, has to join the whole chain first.  GDHIP_BATCH_TEST_SIDE_BINNING_DELAY_MS holds
    the chain's second half back (without the join: 'map::at' from the class table, seen once with a slower copy path)."""
    recipe = synth.config_c3(2_000_000, 50)
    pairs = synth.triangle_pairs(50)
    monkeypatch.setenv("GETDIST_AMD_NATIVE_BATCH", "1")
    ref = mc_of(recipe).get2DDensities(pairs)
    ref[-1].P
    monkeypatch.setenv("GDHIP_BATCH_TEST_SIDE_BINNING_DELAY_MS", "40")
    slow = mc_of(recipe).get2DDensities(pairs)
    same(slow, ref)
    assert any(d.bandwidth_branch == "B" and d.P.shape[0] != 256 for d in slow), "no rule-of-thumb pair in an up-scaled class"


def test_two_objects_from_two_threads_share_a_device(monkeypatch):
    """Two sample sets on one device, each driven from its own host thread (three library threads and two streams
    each): the grids equal those of the same calls made one after the other."""
    monkeypatch.setenv("GETDIST_AMD_NATIVE_BATCH", "1")
    # (unit weights: the real-weight binning kernels add with fp64 atomics, equal to rounding only from run to run)
    recipes = [synth.block_recipe(20, 300_000, weighted=False, stream=64), synth.block_recipe(20, 250_000, weighted=False, stream=65)]
    pairs = synth.triangle_pairs(20)
    serial = [mc_of(r).get2DDensities(pairs) for r in recipes]
    serial = [[d.P.copy() for d in ds] for ds in serial]
    objs = [mc_of(r) for r in recipes]
    out, errs = [None, None], []

    def work(q):
        try:
            for _ in range(3):
                objs[q].ctx.batch2d_invalidate()
                for p in objs[q].paramNames.names:
                    p.N_eff_kde = None
                out[q] = objs[q].get2DDensities(pairs)
                out[q][-1].P
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(q,)) for q in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for q in (0, 1):
        for d, ref in zip(out[q], serial[q]):
            assert np.array_equal(d.P, ref)


def test_mask_moments_evaluated_in_their_consumers_equal_the_array_form(monkeypatch):
    """The prior-mask moments of the boundary / bias corrections (mcsamples.py:1905-1976) are evaluated inside k_boundary
    and k_rows_inv from the windows' summed-area tables; GDHIP_CONV_MOMENT_ARRAYS=1 restores the F x F moment arrays
    (k_mask_eval): the same operations in the same order, so the grids must be equal bit for bit -- for every combination
    of the two correction orders, bounded and unbounded pairs, up-scaled grids."""
    recipe = synth.block_recipe(30, 300_000, weighted=False, stream=66)
    pairs = synth.triangle_pairs(30)
    for kw in (dict(), dict(boundary_correction_order=0), dict(mult_bias_correction_order=0),
               dict(boundary_correction_order=-1, mult_bias_correction_order=2)):
        monkeypatch.setenv("GDHIP_CONV_MOMENT_ARRAYS", "1")
        ref = mc_of(recipe).get2DDensities(pairs, **kw)
        ref = [d.P.copy() for d in ref]
        monkeypatch.delenv("GDHIP_CONV_MOMENT_ARRAYS")
        got = mc_of(recipe).get2DDensities(pairs, **kw)
        assert any(p.has_limits for p in mc_of(recipe).paramNames.names) or True
        for k, (a, b) in enumerate(zip(got, ref)):
            assert np.array_equal(a.P, b), (kw, k, float(np.max(np.abs(a.P - b))))


def test_periodic_pairs_in_both_orientations_repeatedly(monkeypatch, zoo):
    """A periodic parameter as x of one pair and as y of the next (getdist_test.py:181-225's angle / radius set): the two
    pairs cannot share a convolution batch (circular along different axes), so each is gathered out of the class's
    histogram block before its rocFFT frames -- and gd_gather_items once kept its index list at the head of the very
    block it gathered into: the second pair came out as noise in a few percent of the calls.  Forty calls of either
    grid size, every grid `array_equal` to the Python-planned route's."""
    from getdist_amd.mcsamples import MCSamples

    fx = zoo["periodic"]
    build = lambda: MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for kw in fx["kw2"]:
        for pairs in (fx["pairs"], fx["pairs"][::-1]):
            native, plain, mc = both_routes(monkeypatch, build, lambda m: m.get2DDensities(pairs, **kw))
            same(native, plain)
            for rep in range(40):
                again = mc.get2DDensities(pairs, **kw)
                for a, b in zip(again, plain):
                    assert np.array_equal(a.P, b.P), (kw, pairs, rep, float(np.max(np.abs(a.P - b.P))))


@pytest.mark.parametrize("kw", [dict(), dict(smooth_scale_1D=0.3), dict(smooth_scale_1D=1.5), dict(boundary_correction_order=2),
                                dict(mult_bias_correction_order=0, boundary_correction_order=0), dict(fine_bins=512)])
def test_native_1d_entry_equals_python_planned_sequence(monkeypatch, kw):
    """gd_density1d_batch (include/gdhip.h; mcsamples.py:1500-1686 for all parameters in one call) against the
    Python-planned sequence of gd_hist1d / gd_isj1d / gd_density1d over the same kernels: densities, kde_h and N_eff
    `array_equal`; mean-likelihood profiles through the histograms the entry hands back."""
    recipe = synth.block_recipe(12, 300_000, weighted=False, stream=71)

    def call(m):
        return m.get1DDensities(**kw), [p.kde_h for p in m.paramNames.names], [p.N_eff_kde for p in m.paramNames.names]

    (native, h_n, neff_n), (plain, h_p, neff_p), _ = both_routes(monkeypatch, lambda: mc_of(recipe), call)
    assert len(native) == 12
    for a, b in zip(native, plain):
        assert np.array_equal(a.x, b.x) and np.array_equal(a.P, b.P)
    if kw.get("smooth_scale_1D", -1.0) <= 0:
        assert h_n == h_p and neff_n == neff_p and all(h is not None for h in h_n)
    # integer multiplicities and mean likelihoods; real weights (fp64 LDS atomics in the binning: rounding differs run to run)
    s, w, names, ranges = synth.block_recipe(8, 200_000, weighted=True, stream=72)
    loglikes = np.random.default_rng(3).chisquare(4, size=len(s)) / 2
    for weights, tol in ((np.floor(w * 3) + 1.0, 0.0), (w, 1e-9)):
        def build():
            from getdist_amd.mcsamples import MCSamples

            return MCSamples(samples=s, weights=weights, names=names, ranges=ranges, loglikes=loglikes)

        native, plain, _ = both_routes(monkeypatch, build, lambda m: m.get1DDensities([5, 0, 3], meanlikes=True, **kw))
        for a, b in zip(native, plain):
            assert np.max(np.abs(a.P - b.P)) <= tol and np.max(np.abs(a.likes - b.likes)) <= max(tol, 1e-12)
