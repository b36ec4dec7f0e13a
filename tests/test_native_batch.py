"""
CPU tests of the native batch entry's HOST side (getdist_amd/csrc/batch2d.hpp: the plan of a pair batch and the
choreography of the device work), compiled with g++ and driven against the numpy context double: the grids, bandwidths
and optimiser records of MCSamples.get2DDensities through gd_density2d_batch's code must equal those of the
Python-planned route bit for bit, for every branch / grid class / setting the route serves.  (On the GPU the same
comparison runs against the real kernels: tests/test_gpu_native_batch.py.)
"""

import ctypes as C

import numpy as np
import pytest

import native_batch_util as nb


def make(fx, factory, **kw):
    from getdist_amd.mcsamples import MCSamples

    return MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"],
                     _context_factory=factory, **kw)


def same(native, plain):
    assert len(native) == len(plain)
    for k, (a, b) in enumerate(zip(native, plain)):
        assert a.P.shape == b.P.shape, k
        assert np.array_equal(a.P, b.P), (k, float(np.max(np.abs(a.P - b.P))))
        assert np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y) and a.spacing == b.spacing, k
        assert a.bandwidth_branch == b.bandwidth_branch, k
        assert (a.bandwidth is None) == (b.bandwidth is None) and (a.bandwidth is None or a.bandwidth == b.bandwidth), k
        assert (a.kopt is None) == (b.kopt is None), k
        assert a.kopt is None or np.array_equal(a.kopt, b.kopt, equal_nan=True), k
        assert a.view_ranges == b.view_ranges


def test_cholesky_shear_equals_numpy_lapack():
    """np.linalg.cholesky / inv of the 2 x 2 covariance (mcsamples.py:1352-1356) spelled out, pivoting case included."""
    lib = nb.harness()
    rng = np.random.default_rng(5)
    S, r = np.zeros(4), np.zeros(2)
    pivoted = 0
    for t in range(20000):
        a = rng.normal(size=(2, 3)) * 10 ** rng.uniform(-3, 3, size=(2, 1))
        cov = a @ a.T
        Sref = np.linalg.cholesky(cov)
        ich = np.linalg.inv(Sref)
        pivoted += abs(Sref[1, 0]) > abs(Sref[0, 0])
        Sref = Sref * ich[0, 0]
        rref = ich[1, :] / ich[0, 0]
        assert lib.gdt_chol_shear(cov[0, 0], cov[1, 0], cov[1, 1], S.ctypes.data_as(nb._pd), r.ctypes.data_as(nb._pd)) == 0
        assert np.array_equal(S.reshape(2, 2), Sref) and np.array_equal(r, rref), (t, cov)
    assert pivoted > 1000
    assert lib.gdt_chol_shear(1.0, 2.0, 1.0, S.ctypes.data_as(nb._pd), r.ctypes.data_as(nb._pd)) == 1  # not positive definite


def test_scalar_power_is_libm_pow():
    """Python's ``x ** y`` on floats is libm pow; the native plan must not turn pow(x, 2.0) into x * x."""
    lib = nb.harness()
    rng = np.random.default_rng(7)
    for x in (rng.uniform(0.1, 1e7, size=20000)).tolist():
        assert lib.gdt_py_pow(x, 2.0) == x ** 2.0 and lib.gdt_py_pow(x, 1.0 / 6) == x ** (1.0 / 6)


def test_native_route_equals_python_route_block50(zoo):
    """block50: branches A / B / C, four grid sizes, bounded and unbounded pairs; u8 and u16 binning; one and two streams."""
    fx = zoo["block50"]
    plain = make(fx, nb.PlainContext).get2DDensities(fx["pairs"])
    nb.CALLS.clear()
    mc = make(fx, nb.HarnessContext)
    native = mc.get2DDensities(fx["pairs"])
    same(native, plain)
    assert {d.bandwidth_branch for d in native} == {"A", "B", "C"} and len({d.P.shape[0] for d in native}) == 4
    ops = [c[0] for c in nb.CALLS]
    assert "density2d_enqueue" in ops and "kopt2d" in ops and "hist2d_sheared" in ops
    # a second call finds the index columns made and the effective sample numbers known: same grids
    nb.CALLS.clear()
    again = mc.get2DDensities(fx["pairs"])
    same(again, plain)
    assert not any(c[0] in ("prebin8_batch", "autocov_lags_batch", "kde_lag_sums_batch") for c in nb.CALLS)
    mc.ctx.batch2d_invalidate()
    nb.CALLS.clear()
    mc.get2DDensities(fx["pairs"][:5])
    assert any(c[0] == "hist2d_prebinned" for c in nb.CALLS)


def triangle(n):
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def test_native_route_large_call_two_streams_pipelined(zoo):
    """78 pairs with the thresholds lowered so that the call takes the routes of a full triangle: byte-index binning on
    the second context beside the N_eff kernels on the first, the optimiser's launch cut in two, the first part convolved
    on the second stream, side classes."""
    fx = zoo["block50"]
    pairs = triangle(13)
    ref = make(fx, nb.PlainContext)
    plain = ref.get2DDensities(pairs)
    mc = make(fx, nb.HarnessContext)
    mc.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc.KOPT_SPLIT_MIN = 8
    nb.CALLS.clear()
    native = mc.get2DDensities(pairs)
    same(native, plain)
    lanes = {c[1] for c in nb.CALLS if c[0] == "density2d_enqueue"}
    assert len(lanes) == 2, "the convolution used one stream only"
    assert sum(1 for c in nb.CALLS if c[0] == "kopt2d_enqueue") >= 2 and not any(c[0] == "kopt2d" for c in nb.CALLS)
    main, twin = mc.ctx.lane, mc._twin.ctx.lane
    aux = {c[1] for c in nb.CALLS if c[0] == "kopt2d_finish"}
    assert len(aux) == 1 and not aux & {main, twin}
    # N_eff and the optimiser's stage A on the first context, the base grid's binning on the second, the shear chain and
    # get_h on the third
    assert all(c[1] == main for c in nb.CALLS if c[0] in ("kde_lag_sums_batch", "autocov_lags_batch", "kopt2d_enqueue"))
    assert all(c[1] == twin for c in nb.CALLS if c[0] in ("prebin8_batch", "hist2d_prebinned8", "hist2d_prebinned"))
    assert all(c[1] in aux for c in nb.CALLS if c[0] in ("minmax_affine", "hist2d_sheared"))
    assert any(c[0] == "hist2d_prebinned8" for c in nb.CALLS)
    assert [p.N_eff_kde for p in mc.paramNames.names[:13]] == [p.N_eff_kde for p in ref.paramNames.names[:13]]


def test_results_in_flight_tells_the_library_about_a_stream_of_calls(zoo):
    """settings.results_in_flight (end of round 6): 1 when the previous batched call's results have not been waited for -- the
    library then schedules for throughput (the deferred chain beside the first convolution) --, 0 once they have been read:
    the delivery of this call's first grids is what counts then.  Same grids either way."""
    fx = zoo["block50"]
    pairs = triangle(6)
    mc = make(fx, nb.HarnessContext)
    nb.CALLS.clear()
    first = mc.get2DDensities(pairs)          # nothing before it
    second = mc.get2DDensities(pairs)         # the first call's results are still pending
    first[0].P, second[-1].P                  # delivered
    third = mc.get2DDensities(pairs)
    flags = [c[3] for c in nb.CALLS if c[0] == "density2d_batch"]
    assert flags == [0, 1, 0], flags
    same(second, first)
    same(third, first)


def test_native_route_deferred_shear_chain(zoo, monkeypatch):
    """Round 6 (single-triangle latency): in a large call the shear chain and the up-scaled grid classes run on a FOURTH
    context, start when the main class has been binned, the sheared pairs ride in the LAST optimiser part and the main thread
    joins the chain only when it builds that part -- the first part's grids (and their copies) leave earlier.  Same grids,
    bandwidths and records as the planned route; the chain on its own context; stage A of the first part enqueued before the
    chain's last launch returned is not asserted (host timing), the structure is."""
    fx = zoo["block50"]
    pairs = triangle(13)
    ref = make(fx, nb.PlainContext)
    plain = ref.get2DDensities(pairs)
    monkeypatch.setenv("GDHIP_BATCH_SHEAR_DEFERRED_MIN", "16")
    mc = make(fx, nb.HarnessContext)
    mc.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc.KOPT_SPLIT_MIN = 8
    nb.CALLS.clear()
    native = mc.get2DDensities(pairs)
    same(native, plain)
    main, twin = mc.ctx.lane, mc._twin.ctx.lane
    get_h_ctx = {c[1] for c in nb.CALLS if c[0] == "kopt2d_finish"}
    chain_ctx = {c[1] for c in nb.CALLS if c[0] in ("minmax_affine", "hist2d_sheared")}
    assert len(chain_ctx) == 1 and not chain_ctx & ({main, twin} | get_h_ctx), "the deferred chain must have a context of its own"
    assert any(c[0] == "hist2d_sheared" for c in nb.CALLS)
    # the sheared pairs ride in the last base part: the first stage-A launch holds base pairs only
    order = [c[0] for c in nb.CALLS if c[0] in ("kopt2d_enqueue", "hist2d_sheared")]
    assert order.count("kopt2d_enqueue") >= 2
    # the main class is binned in two launches: the first optimiser part's rows, then the others (same context)
    bins = [c for c in nb.CALLS if c[0] == "hist2d_prebinned8"]
    first_part = next(c for c in nb.CALLS if c[0] == "kopt2d_enqueue")
    assert len(bins) == 2 and bins[0][1] == bins[1][1] == twin and bins[0][2] == first_part[2], (bins, first_part)
    monkeypatch.setenv("GDHIP_BATCH_ONE_BINNING", "1")
    mc1 = make(fx, nb.HarnessContext)
    mc1.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc1.KOPT_SPLIT_MIN = 8
    nb.CALLS.clear()
    same(mc1.get2DDensities(pairs), plain)
    assert len([c for c in nb.CALLS if c[0] == "hist2d_prebinned8"]) == 1
    monkeypatch.delenv("GDHIP_BATCH_ONE_BINNING")
    # the switch restores the round-5 order (joined in front of the optimiser, sheared pairs in the first part): same results
    monkeypatch.setenv("GDHIP_BATCH_SHEAR_DEFERRED", "0")
    mc2 = make(fx, nb.HarnessContext)
    mc2.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc2.KOPT_SPLIT_MIN = 8
    same(mc2.get2DDensities(pairs), plain)


@pytest.mark.parametrize("kw", [dict(mult_bias_correction_order=0), dict(boundary_correction_order=0),
                                dict(boundary_correction_order=-1, mult_bias_correction_order=2),
                                dict(smooth_scale_2D=0.5), dict(smooth_scale_2D=2.5), dict(fine_bins_2D=128)])
def test_native_route_settings(zoo, kw):
    fx = zoo["block10_weighted"]
    pairs = triangle(6) + [(7, 2), (9, 8)]
    plain = make(fx, nb.PlainContext).get2DDensities(pairs, **kw)
    native = make(fx, nb.HarnessContext).get2DDensities(pairs, **kw)
    same(native, plain)


def test_native_route_contour_levels_and_names(zoo):
    fx = zoo["c1_bounded"]
    names = fx["names"]
    pairs = [(names[0], names[1]), (names[2], names[3]), (names[3], names[1])]
    plain = make(fx, nb.PlainContext).get2DDensities(pairs, get_density=False, num_plot_contours=2)
    native = make(fx, nb.HarnessContext).get2DDensities(pairs, get_density=False, num_plot_contours=2)
    same(native, plain)
    for a, b in zip(native, plain):
        assert np.array_equal(a.contours, b.contours)


def test_native_route_periodic(zoo):
    fx = zoo["periodic"]
    pairs = triangle(len(fx["names"]))
    plain = make(fx, nb.PlainContext).get2DDensities(pairs)
    native = make(fx, nb.HarnessContext).get2DDensities(pairs)
    same(native, plain)


def test_native_exchange_is_unconditional_and_owned_parameters_only(zoo):
    """Multi-rank N_eff share through the native entry: this rank computes the parameters it owns, the callback -- called
    exactly once, also for a call that needs nothing from it -- delivers the others."""
    from getdist_amd import parallel

    fx = zoo["block50"]
    pairs = triangle(8)
    ref = make(fx, nb.PlainContext)
    plain = ref.get2DDensities(pairs)
    truth = [p.N_eff_kde for p in ref.paramNames.names]
    mc = make(fx, nb.HarnessContext)
    calls = []

    def exchange(mc_):
        calls.append([p.N_eff_kde for p in mc_.paramNames.names[:8]])
        for j in (1, 3, 5, 7):  # what the other rank owns
            mc_.paramNames.names[j].N_eff_kde = truth[j]

    mc._neff_share = parallel.NeffShare([0, 2, 4, 6], exchange)
    nb.CALLS.clear()
    native = mc.get2DDensities(pairs)
    assert len(calls) == 1 and mc._neff_share.exchanged
    assert all(calls[0][j] is not None for j in (0, 2, 4, 6)) and all(calls[0][j] is None for j in (1, 3, 5, 7))
    same(native, plain)
    # a call that needs no N_eff from anybody still enters the collective once
    mc2 = make(fx, nb.HarnessContext)
    calls.clear()
    mc2._neff_share = parallel.NeffShare([0, 1], exchange)
    mc2.get2DDensities([(0, 1)], smooth_scale_2D=0.3)
    assert len(calls) == 1
    # ... and so does a rank without pairs
    mc3 = make(fx, nb.HarnessContext)
    calls.clear()
    mc3._neff_share = parallel.NeffShare([], exchange)
    assert mc3.get2DDensities([]) == [] and len(calls) == 1


def test_native_exchange_over_the_library_communicator(zoo):
    """settings.comm_exchange: the entry exchanges the N_eff values itself -- one sum all-reduce of n doubles, this rank
    contributing the parameters it owns -- and parallel.allgather_neff(comm=...) issues the very same collective, so a
    rank on the Python route and a rank on the native route meet in it."""
    from getdist_amd import parallel

    fx = zoo["block50"]
    pairs = triangle(8)
    ref = make(fx, nb.PlainContext)
    plain = ref.get2DDensities(pairs)
    truth = np.array([0.0 if p.N_eff_kde is None else p.N_eff_kde for p in ref.paramNames.names])
    mc = make(fx, nb.HarnessContext)
    seen = []

    def allreduce(v):  # the other rank owns 1, 3, 5, 7
        seen.append(v.copy())
        other = np.zeros_like(v)
        other[[1, 3, 5, 7]] = truth[[1, 3, 5, 7]]
        return v + other

    mc.ctx.comm_world, mc.ctx.comm_allreduce_sum = 2, allreduce
    share = parallel.NeffShare([0, 2, 4, 6], lambda mc_: (_ for _ in ()).throw(AssertionError("the Python exchange was used")))
    share.library_comm = True
    mc._neff_share = share
    native = mc.get2DDensities(pairs)
    assert len(seen) == 1 and share.exchanged and len(seen[0]) == mc.n
    assert np.all(seen[0][[0, 2, 4, 6]] == truth[[0, 2, 4, 6]]) and np.all(seen[0][[1, 3, 5, 7]] == 0)
    same(native, plain)
    # the Python-level form of the same collective
    mc2 = make(fx, nb.PlainContext)
    for j in (0, 2):
        mc2.paramNames.names[j].N_eff_kde = truth[j]
    seen.clear()

    class Comm:
        world = 2
        allreduce_sum = staticmethod(allreduce)

    parallel.allgather_neff(mc2, [0, 2, 4], mc2.n, comm=Comm())
    assert len(seen) == 1 and seen[0][0] == truth[0] and seen[0][4] == 0 and mc2.paramNames.names[3].N_eff_kde == truth[3]


def test_native_route_correlated_chain_falls_back_to_the_long_route():
    """A chain whose correlation outlasts the 8-lag probe: the entry asks for that N_eff (GD_BATCH2D_NEED_NEFF), the host
    computes it by getCorrelationLength's long route, the second call succeeds; same grids as the Python-planned route."""
    from getdist_amd.mcsamples import MCSamples

    rng = np.random.default_rng(11)
    N = 6000
    e = rng.normal(size=(N, 3))
    x = np.zeros((N, 3))
    for i in range(1, N):
        x[i] = 0.97 * x[i - 1] + e[i]
    kw = dict(samples=x, names=["a", "b", "c"])
    plain = MCSamples(_context_factory=nb.PlainContext, **kw).get2DDensities(triangle(3))
    native = MCSamples(_context_factory=nb.HarnessContext, **kw).get2DDensities(triangle(3))
    same(native, plain)


def test_library_exchange_entered_by_a_call_that_fails_afterwards_is_not_repeated():
    """Advisor finding of round 4: on the library-communicator route the entry may have ENTERED the step's all-reduce before
    it fails (here: GD_BATCH2D_NEED_NEFF from a correlated column nobody owns, raised by the completion step that follows
    the collective).  The library counts the collectives it issued (gd_batch2d_exchanges); the host sets share.exchanged
    from that count before the retry, so the retry and NeffShare.complete() issue no second all-reduce that no other rank
    would match, and the other rank's values -- delivered by the first call -- are kept."""
    from getdist_amd import parallel
    from getdist_amd.mcsamples import MCSamples

    rng = np.random.default_rng(11)
    N = 6000
    e = rng.normal(size=(N, 3))
    x = np.zeros((N, 3))
    for i in range(1, N):
        x[i] = 0.97 * x[i - 1] + e[i]
    kw = dict(samples=x, names=["a", "b", "c"])
    ref = MCSamples(_context_factory=nb.PlainContext, **kw)
    plain = ref.get2DDensities(triangle(3))
    truth = np.array([p.N_eff_kde for p in ref.paramNames.names])
    mc = MCSamples(_context_factory=nb.HarnessContext, **kw)
    seen = []

    def allreduce(v):  # the other rank owns column 1; column 2 belongs to nobody
        seen.append(v.copy())
        other = np.zeros_like(v)
        other[1] = truth[1]
        return v + other

    mc.ctx.comm_world, mc.ctx.comm_allreduce_sum = 2, allreduce
    share = parallel.NeffShare([0], lambda mc_: (_ for _ in ()).throw(AssertionError("a second exchange was entered")))
    share.library_comm = True
    mc._neff_share = share
    # column 0 is this rank's own and correlated as well: give it its value so that the first failure comes AFTER the collective
    mc._init_params([0, 1, 2])
    mc.paramNames.names[0].N_eff_kde = float(truth[0])
    native = mc.get2DDensities(triangle(3))
    assert len(seen) == 1 and share.exchanged and mc.ctx.batch2d_exchanges() == 1
    share.complete(mc)  # what bench.one_step's `finally` does: nothing left to enter
    assert len(seen) == 1
    assert mc.paramNames.names[1].N_eff_kde == truth[1]
    same(native, plain)


def test_grid_sizes_entry(zoo):
    fx = zoo["block50"]
    mc = make(fx, nb.HarnessContext)
    dens = mc.get2DDensities(fx["pairs"])
    from getdist_amd import batch2d

    s = batch2d.settings_of(mc, mc.fine_bins_2D, 1, 1, -1.0, False, None)
    F = mc.ctx.batch2d_grid_sizes(s, mc.n, np.ascontiguousarray(mc.getCorrelationMatrix()), np.asarray(fx["pairs"], dtype=np.int32))
    assert F.tolist() == [d.P.shape[0] for d in dens]
    assert C.sizeof(batch2d.ParamState) == 88 and C.sizeof(batch2d.BatchSettings) == 144


def likes_same(native, plain):
    for k, (a, b) in enumerate(zip(native, plain)):
        assert (a.likes is None) == (b.likes is None), k
        assert a.likes is None or np.array_equal(a.likes, b.likes), (k, float(np.max(np.abs(a.likes - b.likes))))
        assert (a.mask is None) == (b.mask is None) and (a.mask is None or np.array_equal(a.mask, b.mask)), k


def test_native_route_serves_injected_bandwidths(zoo):
    """``_bandwidths`` (the parity tests' closed loop: the oracle's triples) is an input of the native call."""
    fx = zoo["block50"]
    pairs = triangle(7)
    rng = np.random.default_rng(3)
    plain_mc, mc = make(fx, nb.PlainContext), make(fx, nb.HarnessContext)
    auto = plain_mc.get2DDensities(pairs)
    triples = [(d.bandwidth[0] * rng.uniform(0.8, 1.3), d.bandwidth[1] * rng.uniform(0.8, 1.3), 0.9 * d.bandwidth[2]) for d in auto]
    plain = plain_mc.get2DDensities(pairs, _bandwidths=triples)
    nb.CALLS.clear()
    native = mc.get2DDensities(pairs, _bandwidths=triples)
    same(native, plain)
    assert not any(c[0].startswith("kopt2d") for c in nb.CALLS), "no optimiser launch with injected bandwidths"
    assert [d.bandwidth for d in native] == [tuple(t) for t in triples]


def test_native_route_serves_2d_effective_samples():
    # a correlated chain (AR(1) with different memory per parameter): the 2D estimate differs from the smaller 1D one
    rng = np.random.default_rng(8)
    N, rho = 20000, np.array([0.6, 0.3, 0.0, 0.8])
    e = rng.standard_normal((N, 4))
    x = np.zeros((N, 4))
    for t in range(1, N):
        x[t] = rho * x[t - 1] + e[t]
    x[:, 1] += 0.5 * x[:, 0]
    fx = dict(samples=x, weights=None, names=["a", "b", "c", "d"], ranges={})
    pairs = triangle(4)
    plain_mc, mc = make(fx, nb.PlainContext), make(fx, nb.HarnessContext)
    plain_mc.use_effective_samples_2D = mc.use_effective_samples_2D = True
    plain = plain_mc.get2DDensities(pairs)
    nb.CALLS.clear()
    native = mc.get2DDensities(pairs)
    same(native, plain)
    assert any(c[0] == "density2d_enqueue" for c in nb.CALLS)
    ordinary = make(fx, nb.HarnessContext).get2DDensities(pairs)
    assert any(a.bandwidth != b.bandwidth for a, b in zip(native, ordinary)), "the 2D estimate must have been used"


def test_native_route_serves_mean_likelihoods(zoo):
    fx = zoo["block10_weighted"]
    pairs = triangle(5)
    from oracle.fixtures import loglikes_for

    ll = loglikes_for(fx["samples"])
    plain = make(fx, nb.PlainContext, loglikes=ll).get2DDensities(pairs, meanlikes=True)
    native = make(fx, nb.HarnessContext, loglikes=ll).get2DDensities(pairs, meanlikes=True)
    same(native, plain)
    likes_same(native, plain)
    assert all(d.likes is not None for d in native)


@pytest.mark.parametrize("meanlikes", [False, True])
def test_native_route_serves_a_mask_callback(zoo, meanlikes):
    fx = zoo["c1_bounded"]
    names = fx["names"]
    pairs = [(names[0], names[1]), (names[2], names[3]), (names[3], names[1])]

    def mask_function(minx, miny, stepx, stepy, mask):
        ny, nx = mask.shape
        x = minx + stepx * np.arange(nx)
        y = miny + stepy * np.arange(ny)
        mask[(y[:, None] - y[ny // 2]) > 1.5 * (x[None, :] - x[nx // 2]) + 10 * stepy] = 0

    from oracle.fixtures import loglikes_for

    ll = loglikes_for(fx["samples"])
    plain = make(fx, nb.PlainContext, loglikes=ll).get2DDensities(pairs, mask_function=mask_function, meanlikes=meanlikes)
    nb.CALLS.clear()
    native = make(fx, nb.HarnessContext, loglikes=ll).get2DDensities(pairs, mask_function=mask_function, meanlikes=meanlikes)
    same(native, plain)
    likes_same(native, plain)
    assert all(d.mask is not None and d.mask.any() for d in native)
    assert not any(c[0] == "density2d_enqueue" for c in nb.CALLS), "bandwidths_only: the call itself convolves nothing"
    levels_plain = make(fx, nb.PlainContext).get2DDensities(pairs, mask_function=mask_function, get_density=False, num_plot_contours=2)
    levels_native = make(fx, nb.HarnessContext).get2DDensities(pairs, mask_function=mask_function, get_density=False, num_plot_contours=2)
    for a, b in zip(levels_native, levels_plain):
        assert np.array_equal(a.contours, b.contours)


def test_native_route_second_binning_launch_wraps_a_counter(monkeypatch):
    """The main binning runs in two launches (the first optimiser part's rows first).  A 16-bit counter that wraps in the
    SECOND launch -- a column with 60 % of its samples on one value, paired last -- sends the whole class through the u16
    redo while the first part is already being optimised on rows of the first buffer: that buffer stays alive until the call
    ends, the later parts and every convolution read the redone one; same grids as the planned route."""
    rng = np.random.default_rng(12)
    N = 150_000
    x = rng.standard_normal((N, 12))
    spike = rng.random(N) < 0.7  # the last two columns share it: their pair -- the triangle's last -- has a bin with 0.7 N samples
    x[:, 10] = np.where(spike, -0.5, x[:, 10])
    x[:, 11] = np.where(spike, 0.25, x[:, 11])
    fx = dict(samples=x, weights=None, names=["q%d" % i for i in range(12)], ranges={})
    pairs = triangle(12)
    assert pairs[-1] == (10, 11)
    plain = make(fx, nb.PlainContext).get2DDensities(pairs)
    monkeypatch.setenv("GDHIP_BATCH_SHEAR_DEFERRED_MIN", "16")
    mc = make(fx, nb.HarnessContext)
    mc.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc.KOPT_SPLIT_MIN = 8
    nb.CALLS.clear()
    native = mc.get2DDensities(pairs)
    same(native, plain)
    byte_launches = [c for c in nb.CALLS if c[0] == "hist2d_prebinned8"]
    assert len(byte_launches) == 2 and byte_launches[0][2] < len(pairs), byte_launches
    redo = [c for c in nb.CALLS if c[0] == "hist2d_prebinned" and c[3] == 256]
    assert len(redo) == 1 and redo[0][2] == len(pairs), (redo, "the second launch's wrap sends the whole class through the u16 redo")
