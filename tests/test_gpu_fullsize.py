"""
GPU tests at BASELINE.json's full row count (N = 1e7) through size-independent properties, plus the property tests
the reference itself uses (getdist/tests/getdist_test.py:144-165 mirror symmetry, :227-238 pooled mean).
"""

import numpy as np
import pytest

from getdist_amd import synth

pytestmark = pytest.mark.gpu

N_FULL = 10_000_000


@pytest.fixture(scope="module")
def big():
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = synth.block_recipe(10, N_FULL, weighted=False, stream=31)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    return mc, s


def test_moments_and_quantile_rank_property(big):
    mc, s = big
    assert np.array_equal(mc._col_min, s.min(axis=0)) and np.array_equal(mc._col_max, s.max(axis=0))
    assert np.allclose(mc.means, s.mean(axis=0), rtol=1e-11, atol=1e-13)
    fr = np.array([0.001, 0.1, 0.5, 0.9, 0.999])
    for j in (0, 4, 9):
        q = mc.confidence(j, fr)
        x = s[:, j]
        for f, v in zip(fr, q):
            target = mc.norm * f
            assert np.sum(x <= v) >= target  # cumulative weight reaches the target at v ...
            assert np.sum(x < v) < target    # ... and not before (chains.py:836 searchsorted-left semantics)
            assert np.any(x == v)            # a sample value, not an interpolation


def test_histogram_properties(big):
    mc, s = big
    mc.prepareParams(neff=False)
    F = 256
    names = mc.paramNames.names
    pairs = [(0, 1), (5, 6), (4, 9), (8, 9)]
    e = {j: mc._bin_edges(names[j], F) for j in {p for pr in pairs for p in pr}}
    ctx = mc.ctx
    idx = {j: mc._index_column(j, F, e[j][1], e[j][0]) for j in e}
    Hp = ctx.hist2d_prebinned([idx[a] for a, b in pairs], [idx[b] for a, b in pairs], F).to_host((len(pairs), F, F))
    Hd = ctx.hist2d([a for a, b in pairs], [b for a, b in pairs], [e[a][1] for a, b in pairs], [e[a][0] for a, b in pairs],
                    [e[b][1] for a, b in pairs], [e[b][0] for a, b in pairs], F).to_host((len(pairs), F, F))
    assert np.array_equal(Hp, Hd)  # pre-binned and fused-fp64 kernels agree bit for bit (integer counts)
    h1 = ctx.hist1d(sorted(e), [e[j][1] for j in sorted(e)], [e[j][0] for j in sorted(e)], F)
    for k, (a, b) in enumerate(pairs):
        assert Hp[k].sum() == N_FULL                       # every sample lands in the grid
        assert np.array_equal(Hp[k].sum(axis=0), h1[sorted(e).index(a)])  # marginal over y == 1D histogram of x
        assert np.array_equal(Hp[k].sum(axis=1), h1[sorted(e).index(b)])
    ixs = ((s[:, 0] - e[0][1]) / e[0][0] + 0.5).astype(int)  # one column against numpy at full size
    assert np.array_equal(h1[sorted(e).index(0)], np.bincount(ixs, minlength=F))


def test_full_size_densities_are_sane_and_deterministic(big):
    mc, s = big
    pairs = [(0, 1), (5, 6), (4, 9)]
    d1 = mc.get2DDensities(pairs)
    d2 = mc.get2DDensities(pairs)
    for a, b in zip(d1, d2):
        assert a.P.max() == 1.0 and a.P.min() > -1e-12
        assert np.array_equal(a.P, b.P)  # the same call twice on one device: every reduction has a fixed order
    p = mc.get1DDensities([0, 4, 9])
    for d in p:
        assert d.P.max() == 1.0 and d.P.shape == (1024,)


def test_c3_full_shape_determinism_and_one_pair_per_census_class_against_the_oracle(tmp_path):
    """C3 as BASELINE.json states it -- 50 parameters x 1e7 rows, all 1225 pairs: two identical calls on one device give
    bit-identical grids (every reduction has a fixed order), and one pair of each (bandwidth branch, #bounded, grid size)
    class is compared with the oracle at full size (strict 1e-6 gate; a TNC pair above it must be chaotic in the oracle
    and inside the oracle's own spread)."""
    import multiprocessing as mp
    import os
    import shutil
    import tempfile

    from getdist_amd.mcsamples import MCSamples
    from parity_workers import oracle_pair

    s, w, names, ranges = synth.config_c3(N_FULL, 50)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    pairs = synth.triangle_pairs(50)
    dens = mc.get2DDensities(pairs)
    par = mc.paramNames.names
    again = mc.get2DDensities(pairs)
    differing = [pr for pr, d1, d2 in zip(pairs, dens, again) if not np.array_equal(d1.P, d2.P)]
    assert not differing, ("same call, same device, different grids", len(differing), differing[:5])
    # one rank's share of an 8-rank job (153 pairs, dealt like bench.py does): its batches are convolved alternately on
    # the streams of two contexts, in other batch compositions than above -- the grids must not notice
    share = pairs[3::8]
    lo, hi = mc.CONV_TWO_STREAMS_PAIRS
    assert lo <= len(share) <= hi
    part = mc.get2DDensities(share)
    full = dict(zip(pairs, dens))
    differing = [pr for pr, d in zip(share, part) if not np.array_equal(d.P, full[pr].P)]
    assert not differing, ("a share of the triangle differs from the same pairs of the full call", len(differing), differing[:5])
    klass = {}
    for (a, b), d in zip(pairs, dens):
        assert d.P.max() == 1.0 and d.P.min() > -1e-12
        key = "%s/%d/%d" % (d.bandwidth_branch, int(bool(par[a].has_limits)) + int(bool(par[b].has_limits)), d.P.shape[0])
        klass.setdefault(key, []).append((a, b))
    assert len(klass) >= 8 and {256, 384, 768, 960} <= {int(k.split("/")[2]) for k in klass}  # SURVEY 8d census
    picks = [members[len(members) // 2] for _, members in sorted(klass.items())]
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="gdamd_c3_", dir=shm)
    try:
        tasks = []
        for (a, b) in picks:
            path = os.path.join(tmp, "p%d_%d.npy" % (a, b))
            np.save(path, np.ascontiguousarray(s[:, [a, b]]))
            sub = [names[a], names[b]]
            tasks.append(dict(pair=(a, b), path=path, names=sub, ranges={k: v for k, v in ranges.items() if k in sub}))
        with mp.get_context("spawn").Pool(min(len(tasks), os.cpu_count() or 1)) as pool:
            results = pool.map(oracle_pair, tasks, chunksize=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    by_pair = dict(zip(pairs, dens))
    from oracle import kde_oracle as ko

    report = {}
    for r in results:
        d = by_pair[tuple(r["pair"])]
        assert d.bandwidth_branch == r["branch"] and d.P.shape == r["P"].shape, r["pair"]
        err = float(np.max(np.abs(d.P - r["P"])))
        report["%s-%s" % (names[r["pair"][0]], names[r["pair"][1]])] = dict(branch=r["branch"], F=int(d.P.shape[0]), max_abs_dP=err)
        if r["tnc"]:
            assert abs(d.kopt[0] - r["t_star"]) <= 1e-10 * r["t_star"]
            assert np.max(np.abs(np.asarray(d.kopt[1:7]) - r["psi"]) / np.abs(r["psi"])) <= 1e-10
        if err >= 1e-6:
            assert r["tnc"], (r["pair"], err)
            verdict = ko.judge_triple(d.kopt[8:11], r["psi"], r["opt_N"], ensembles=r["ensembles"])
            assert verdict["moved"] > 1e-6, (r["pair"], "oracle stable, device off", err, verdict)
            assert verdict["ok"], (r["pair"], d.kopt[8:11], verdict)
            assert err < 5e-4, (r["pair"], err)
    _report("C3_census_parity", dict(classes=len(klass), pairs_checked=len(results), per_pair=report,
                                     identical_reruns=len(pairs)))
    mc.ctx.close()


def test_mirror_symmetry_like_reference():
    """getdist_test.py:144-165: mirrored samples give mirrored densities, with hard bounds on both axes."""
    from getdist_amd.mcsamples import MCSamples

    r = np.random.default_rng(10)
    n = 1_000_000
    x = np.abs(r.normal(0.4, 0.5, n))
    x = x[x < 1.2][:700_000]
    y = r.normal(0.0, 1.0, len(x))
    y = np.where(np.abs(y) > 1.5, 1.5 * np.sign(y) - (y - 1.5 * np.sign(y)), y)
    s = np.column_stack([x, y])
    a = MCSamples(samples=s, names=["x", "y"], ranges={"x": (0, 1.2), "y": (-1.5, 1.5)})
    b = MCSamples(samples=np.column_stack([1.2 - x, y]), names=["x", "y"], ranges={"x": (0, 1.2), "y": (-1.5, 1.5)})
    c = MCSamples(samples=np.column_stack([x, -y]), names=["x", "y"], ranges={"x": (0, 1.2), "y": (-1.5, 1.5)})
    pa, pb = a.get1DDensity("x").P, b.get1DDensity("x").P
    assert np.allclose(pa, pb[::-1], atol=1e-5)
    da, db, dc = a.get2DDensity("x", "y").P, b.get2DDensity("x", "y").P, c.get2DDensity("x", "y").P
    assert np.allclose(da, db[:, ::-1], atol=1e-5)
    assert np.allclose(da, dc[::-1, :], atol=1e-5)


def test_pooled_mean_of_chain_list():
    """getdist_test.py:227-238"""
    from getdist_amd.mcsamples import MCSamples

    r = np.random.default_rng(3)
    chains = [r.normal(k, 1.0, (20_000 + 1000 * k, 3)) for k in range(3)]
    ws = [r.integers(1, 4, len(c)).astype(float) for c in chains]
    mc = MCSamples(samples=chains, weights=ws, names=["a", "b", "c"])
    allx, allw = np.vstack(chains), np.hstack(ws)
    assert np.allclose(mc.getMeans(), allw.dot(allx) / allw.sum(), rtol=1e-12)
    assert list(mc.chain_offsets) == [0, 20000, 41000, 63000]


def test_weighted_full_size_properties():
    """N = 1e7 with real-valued weights: fp64 LDS atomics, 4 stripes; mass / marginal / variant-agreement properties."""
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = synth.block_recipe(5, N_FULL, weighted=True, stream=32)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    assert abs(mc.norm - np.sum(w)) <= 1e-12 * mc.norm
    assert np.allclose(mc.means, w.dot(s) / np.sum(w), rtol=1e-11, atol=1e-13)
    mc.prepareParams(neff=False)
    F = 256
    par = mc.paramNames.names
    e = [mc._bin_edges(p, F) for p in par]
    ctx = mc.ctx
    pairs = [(0, 1), (3, 4)]
    idx = [mc._index_column(j, F, e[j][1], e[j][0]) for j in range(5)]
    Hp = ctx.hist2d_prebinned([idx[a] for a, b in pairs], [idx[b] for a, b in pairs], F).to_host((2, F, F))
    Hd = ctx.hist2d([a for a, b in pairs], [b for a, b in pairs], [e[a][1] for a, b in pairs], [e[a][0] for a, b in pairs],
                    [e[b][1] for a, b in pairs], [e[b][0] for a, b in pairs], F).to_host((2, F, F))
    h1 = ctx.hist1d(list(range(5)), [x[1] for x in e], [x[0] for x in e], F)
    for k, (a, b) in enumerate(pairs):
        assert abs(Hp[k].sum() - mc.norm) <= 1e-11 * mc.norm
        assert np.allclose(Hp[k], Hd[k], rtol=1e-11, atol=1e-9)
        assert np.allclose(Hp[k].sum(axis=0), h1[a], rtol=1e-11, atol=1e-8)
        assert np.allclose(Hp[k].sum(axis=1), h1[b], rtol=1e-11, atol=1e-8)
    ixs = ((s[:, 0] - e[0][1]) / e[0][0] + 0.5).astype(int)
    assert np.allclose(h1[0], np.bincount(ixs, weights=w, minlength=F), rtol=1e-11, atol=1e-8)
    q = mc.confidence(2, np.array([0.05, 0.5, 0.95]))
    for f, v in zip([0.05, 0.5, 0.95], q):
        below = np.sum(w[s[:, 2] < v])
        upto = np.sum(w[s[:, 2] <= v])
        assert below < mc.norm * f * (1 + 1e-9) and upto >= mc.norm * f * (1 - 1e-9)
    d = mc.get1DDensities([0, 4])
    assert all(x.P.max() == 1.0 for x in d)
    d2 = mc.get2DDensities(pairs)
    assert all(x.P.max() == 1.0 and x.P.shape == (F, F) for x in d2)
    # ... and the finished real-weight 2D grids at full size against the oracle (mcsamples.py:1728: np.bincount with
    # weights, then the whole density path), 1e-6 or the oracle-ensemble criterion for a chaotic TNC pair
    import golden_util as gu
    from oracle import kde_oracle as ko

    for (a, b), dd in zip(pairs, d2):
        orc = ko.OracleSamples(s[:, [a, b]], w, names=[names[a], names[b]], ranges={k: v for k, v in ranges.items() if k in (names[a], names[b])})
        for k in (0, 1):
            orc.init_param(k)
            orc.pars[k].N_eff_kde = par[(a, b)[k]].N_eff_kde  # (checked to 1e-9 elsewhere; its FFT route costs ~10 s per column here)
        tr = {}
        o = orc.density_2d(0, 1, trace=tr)
        err, loose = gu.assert_grid_or_oracle_ensemble(dd, o, tr, "weighted %s-%s" % (names[a], names[b]),
                                                       oracle_at=lambda bw: orc.density_2d(0, 1, _bandwidths=bw)["P"])
        _report("weighted_2d_full_size_%d_%d" % (a, b), dict(max_abs_dP=err, loose=loose, N=N_FULL))


def test_full_size_grids_match_the_oracle(big):
    """N = 1e7: complete 1D and 2D densities against the oracle on the same inputs (pairs whose bandwidth does not
    pass through TNC, so the strict 1e-6 tolerance applies end to end)."""
    from oracle import kde_oracle as ko

    mc, s = big
    names = [p.name for p in mc.paramNames.names]
    ranges = {nm: (mc.ranges.getLower(nm), mc.ranges.getUpper(nm)) for nm in names
              if mc.ranges.getLower(nm) is not None or mc.ranges.getUpper(nm) is not None}
    cols = [4, 5, 9]
    orc = ko.OracleSamples(np.ascontiguousarray(s[:, cols]), names=[names[c] for c in cols],
                           ranges={k: v for k, v in ranges.items() if k in [names[c] for c in cols]})
    d1 = mc.get1DDensities(cols)
    for k in range(3):
        o = orc.density_1d(k)
        assert np.max(np.abs(d1[k].P - o["P"])) < 1e-6, names[cols[k]]
        par = mc.paramNames.names[cols[k]]
        assert abs(par.N_eff_kde - orc.pars[k].N_eff_kde) <= 1e-9 * par.N_eff_kde
    for (a, b) in ((0, 2), (1, 2)):  # (p4,p9) both bounded, (p5,p9) one bounded
        d = mc.get2DDensities([(cols[a], cols[b])])[0]
        tr = {}
        o = orc.density_2d(a, b, trace=tr)
        assert d.bandwidth_branch == tr["branch"]
        assert np.allclose(d.bandwidth, (tr["hx"], tr["hy"], tr["c"]), rtol=1e-6, atol=1e-12)
        assert np.max(np.abs(d.P - o["P"])) < 1e-6, (names[cols[a]], names[cols[b]])
    # unbounded pairs: the bandwidth passes through the device's TNC.  Strict gate wherever the oracle's own map is
    # stable under a 1e-15 perturbation of its inputs; the loose gate only if it is demonstrably chaotic there.
    ucols = [5, 6, 7]
    orc_u = ko.OracleSamples(np.ascontiguousarray(s[:, ucols]), names=[names[c] for c in ucols])
    strict = 0
    for (a, b) in ((0, 1), (1, 2)):  # (p5,p6): correlated block -> sheared branch A; (p6,p7)
        d = mc.get2DDensities([(ucols[a], ucols[b])])[0]
        tr = {}
        o = orc_u.density_2d(a, b, trace=tr)
        assert d.bandwidth_branch == tr["branch"] and not (mc.paramNames.names[ucols[a]].has_limits
                                                           or mc.paramNames.names[ucols[b]].has_limits)
        assert abs(d.kopt[0] - tr["t_star"]) <= 1e-10 * tr["t_star"]
        bw_err = float(np.max(np.abs(np.array(d.bandwidth) - (tr["hx"], tr["hy"], tr["c"]))) / max(tr["hx"], tr["hy"]))
        if bw_err < 1e-6:
            strict += 1
            assert np.max(np.abs(d.P - o["P"])) < 1e-6, (names[ucols[a]], names[ucols[b]])
        else:
            psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
            verdict = ko.judge_triple(d.kopt[8:11], psi, tr["opt_N"], tr["opt_corr"])
            assert verdict["moved"] > 1e-6, (names[ucols[a]], names[ucols[b]], bw_err, verdict)
            assert verdict["ok"], (d.kopt[8:11], verdict)
            assert np.max(np.abs(d.P - o["P"])) < 5e-4
    print("full-size unbounded TNC pairs on the strict gate: %d of 2" % strict)


def test_thinning_at_scale_matches_the_oracle():
    """gd_thin_rows with more scan tiles than one pass of the tile-sum scan holds (carry path), on sub-ranges, both
    branches of chains.py:878-916, against the oracle's cumsum restatement."""
    from getdist_amd.mcsamples import MCSamples
    from oracle import convergence_oracle as co

    rng = np.random.default_rng(11)
    N = 2_600_000  # 1270 tiles of 2048 rows
    w = rng.geometric(0.3, N).astype(np.float64)
    x = rng.standard_normal((N, 1))
    offs = [0, 900_001, N]
    mc = MCSamples(samples=[x[a:b] for a, b in zip(offs[:-1], offs[1:])], weights=[w[a:b] for a, b in zip(offs[:-1], offs[1:])],
                   names=["x"])
    for factor in (3, int(w.max()), int(w.max()) + 5):
        for lo, hi in ((0, N), (offs[1], N), (0, offs[1])):
            buf, K = mc._thin_rows(factor, lo, hi)
            got = buf.to_host((K,), dtype=np.int32).astype(np.int64)
            buf.free()
            want = co.thin_indices(factor, w[lo:hi]) + lo
            assert K == len(want) and np.array_equal(got, want), (factor, lo, hi)


# ---- the other BASELINE.json configurations at their full sizes --------------------------------------------------------
def _report(key, value):
    import json
    import os

    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "r04_configs.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = value
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("n,weighted", [(100, True), (100, False), (200, True), (64, True), (65, False), (230, True), (3, True)])
def test_covariance_multi_tile_against_numpy(n, weighted):
    """gd_cov with more columns than one 64-column tile (C4's n = 100: 3 tile pairs, C5's n = 200: 10) at N > 1e6,
    against the two-pass weighted covariance in numpy (chains.py:709-733); also on row sub-ranges."""
    from getdist_amd._lib import Context

    N = 1_200_037
    r = np.random.default_rng(n)
    A = r.standard_normal((n, n)) / np.sqrt(n) + np.eye(n)
    s = r.standard_normal((N, n)) @ A.T + r.uniform(-5, 5, n)
    w = r.exponential(1.0, N) if weighted else None
    ctx = Context(0)
    ctx.upload(np.asfortranarray(s), w)
    for lo, hi in ((0, N), (100_003, 900_001)):
        means, cov, norm = ctx.cov(None, lo, hi)
        ww = np.ones(hi - lo) if w is None else w[lo:hi]
        x = s[lo:hi]
        m = ww @ x / ww.sum()
        d = x - m
        want = (d * ww[:, None]).T @ d / ww.sum()
        sd = np.sqrt(np.diag(want))
        assert abs(norm - ww.sum()) <= 1e-12 * ww.sum()
        assert np.max(np.abs(means - m) / sd) < 1e-12
        assert np.max(np.abs(cov - want) / np.outer(sd, sd)) < 1e-11, (n, weighted, lo, hi)
        assert np.array_equal(cov, cov.T)
    sub = [3 % n, n - 1, 17 % n, 64 % n, 0] if n > 4 else [2, 0]
    _, cov_sub, _ = ctx.cov(sub)
    assert np.allclose(cov_sub, ctx.cov(None)[1][np.ix_(sub, sub)], rtol=1e-12, atol=0)
    ctx.close()


def test_c4_full_size_gelman_rubin_against_numpy():
    """C4 at its full size -- 8 chains x 5e6 rows x 100 parameters, w ~ Exp(1), all chains on one GPU (32 GB resident):
    Gelman-Rubin eigenvalues and the MeanVar statistics against the same formulas (chains.py:1446-1474,
    mcsamples.py:964-985) evaluated with numpy BLAS on the host, relative 1e-9."""
    import time

    from getdist_amd.mcsamples import MCSamples
    from getdist_amd.parallel import gelman_rubin_from_chain_stats

    nch, N, n = 8, 5_000_000, 100
    chains, ws = [], []
    for c in range(nch):
        s, w, names = synth.config_c4_chain(c, N, n)
        chains.append(s)
        ws.append(w)
    t0 = time.perf_counter()
    mc = MCSamples(samples=chains, weights=ws, names=names)
    t_ctor = time.perf_counter() - t0
    mc.ctx.sync()
    t0 = time.perf_counter()
    D = mc.getGelmanRubinEigenvalues()
    mv = mc.getMeanVarTest()
    t_conv = time.perf_counter() - t0
    mc._chain_stats_cache = {}
    mc.ctx.timer_start()
    mc.getSeparateChainStats(n)
    t_kernels = mc.ctx.timer_stop_ms()
    stats = []
    tot_w = sum(float(w.sum()) for w in ws)
    pooled = sum(w @ s for s, w in zip(chains, ws)) / tot_w
    for s, w in zip(chains, ws):
        m = w @ s / w.sum()
        d = s - m
        stats.append((m, (d * w[:, None]).T @ d / w.sum(), float(w.sum())))
    want = gelman_rubin_from_chain_stats(stats, pooled)
    assert np.allclose(mc.means, pooled, rtol=1e-11)
    # 8 chains give a between-chain matrix of rank 7: the other 93 eigenvalues are rounding noise around zero
    big = np.abs(want) > 1e-9 * np.max(want)
    assert np.sum(big) == nch - 1
    assert np.max(np.abs(D - want)[big] / np.abs(want)[big]) < 1e-9, (D[big], want[big])
    assert np.max(np.abs(D - want)[~big]) < 1e-12 * np.max(want)
    between = sum((m - pooled) ** 2 for m, _, _ in stats) / (nch - 1)
    within = sum(np.diag(c) * nw for _, c, nw in stats) / tot_w
    assert np.allclose(mv, np.sqrt(between / within), rtol=1e-9)
    alg_bytes = nch * 2 * 8.0 * N * (n + 1)
    _report("C4", dict(chains=nch, rows_per_chain=N, params=n, construct_upload_s=round(t_ctor, 2),
                       gelman_rubin_plus_meanvar_ms=round(t_conv * 1e3, 2), chain_covariances_ms=round(t_kernels, 2),
                       algorithmic_GB=round(alg_bytes / 1e9, 2), algorithmic_GBps=round(alg_bytes / t_kernels / 1e6, 1),
                       flops_T=round(nch * N * n * n * 2 / 1e12, 3), TFLOPs=round(nch * N * n * n * 2 / t_kernels / 1e9, 2),
                       GR=float(np.max(D)), max_rel_error_vs_numpy=float(np.max(np.abs(D - want)[big] / np.abs(want)[big]))))


def test_c2_weighted_full_size_1d_grids_against_the_oracle():
    """C2 at its full size -- 30 parameters, N = 1e7, w ~ Exp(1), seven hard-bounded parameters: 1D density grids of a
    bimodal, an unbounded, two one-sided bounded parameters and one of the strongly correlated block against the
    oracle on the same weighted columns, 1e-6 of the peak; N_eff 1e-9; limit flags exact."""
    import time

    from getdist_amd.mcsamples import MCSamples
    from oracle import kde_oracle as ko

    s, w, names, ranges = synth.config_c2(N_FULL)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    mc.get1DDensities()
    times = []
    for _ in range(3):
        for p in mc.paramNames.names:
            p.N_eff_kde = None
            p._ranges_done = False
        mc._initLimits()
        mc.density1D = {}
        mc.ctx.sync()
        t0 = time.perf_counter()
        dens = mc.get1DDensities()
        times.append(time.perf_counter() - t0)
    cols = [0, 7, 4, 24, 22]
    sub = [names[c] for c in cols]
    orc = ko.OracleSamples(np.ascontiguousarray(s[:, cols]), w, names=sub, ranges={k: v for k, v in ranges.items() if k in sub})
    worst = 0.0
    for k, c in enumerate(cols):
        o = orc.density_1d(k)
        par, opar = mc.paramNames.names[c], orc.pars[k]
        assert (bool(par.has_limits_bot), bool(par.has_limits_top)) == (bool(opar.has_limits_bot), bool(opar.has_limits_top))
        assert abs(par.N_eff_kde - opar.N_eff_kde) <= 1e-9 * par.N_eff_kde
        assert abs(par.kde_h - opar.kde_h) <= 1e-8 * opar.kde_h, (names[c], par.kde_h, opar.kde_h)
        err = float(np.max(np.abs(dens[c].P - o["P"])))
        worst = max(worst, err)
        assert err < 1e-6, (names[c], err)
    assert {bool(mc.paramNames.names[c].has_limits) for c in cols} == {True, False}
    _report("C2", dict(params=30, rows=N_FULL, weights="Exp(1) fp64", all_30_densities_ms=round(min(times) * 1e3, 2),
                       densities_per_s=round(30 / min(times), 1), checked_params=sub, max_abs_grid_error_vs_oracle=worst,
                       bounded=[p.name for p in mc.paramNames.names if p.has_limits]))


def _c5_fits():
    """C5 keeps 80 GB of samples on the host and on the device: needs 100 GB of free HBM and 130 GB of free host RAM."""
    try:
        import psutil

        from getdist_amd._lib import Context

        ctx = Context(0)
        free = ctx.device_info()["hbm_free"]
        ctx.close()
        return free > 100e9 and psutil.virtual_memory().available > 130e9
    except Exception:
        return False


def test_c5_stress_full_size_properties():
    """C5 -- 200 parameters x 5e7 rows resident on ONE GPU (80 GB): margestats of all 200 parameters, then the full
    19 900-pair triangle in slabs, through size-independent properties (normalisation, determinism of a re-run pair,
    2D marginal mass = 1D mass, quantile rank property), and one pair against the oracle at full size."""
    import time

    from getdist_amd.mcsamples import MCSamples

    if __import__("os").environ.get("GETDIST_AMD_SKIP_C5", "0") == "1" or not _c5_fits():
        pytest.skip("C5 needs 100 GB of free HBM and 130 GB of free host memory")
    n, N = 200, 50_000_000
    t0 = time.perf_counter()
    s, w, names, ranges = synth.block_recipe(n, N, weighted=False, stream=7)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    t_ctor = time.perf_counter() - t0
    t0 = time.perf_counter()
    ms = mc.getMargeStats()
    t_marge = time.perf_counter() - t0
    for j in (0, 57, 199):
        lim = ms.parWithName(names[j]).limits[0]
        x = s[:, j]
        inside = np.mean((x >= lim.lower) & (x <= lim.upper))
        if lim.twotail:  # the outermost equal-density crossings: exactly 68 % for a unimodal marginal, more when the
            assert inside > 0.67 and (inside < 0.69 or j == 0), (names[j], inside)  # interval spans a valley (p0 is bimodal)
    q = mc.confidence(123, np.array([0.025, 0.5]))
    for f, v in zip((0.025, 0.5), q):
        assert np.sum(s[:, 123] <= v) >= N * f > np.sum(s[:, 123] < v)
    pairs = synth.triangle_pairs(n)
    t0 = time.perf_counter()
    done = 0
    classes = set()
    for a in range(0, len(pairs), 2000):
        dens = mc.get2DDensities(pairs[a:a + 2000])
        for d in dens:
            assert d.P.max() == 1.0 and d.P.min() > -1e-12
            classes.add(int(d.P.shape[0]))
        done += len(dens)
    t_tri = time.perf_counter() - t0
    # the same triangle again: plans (rocFFT's run-time compiled kernels for the 1152- and 1440-point frames of the
    # up-scaled classes: ~0.6 s each), scratch, page-locked result blocks and index columns exist now
    t0 = time.perf_counter()
    for a in range(0, len(pairs), 2000):
        dens = mc.get2DDensities(pairs[a:a + 2000])
        dens[len(dens) - 1].P
    t_tri_warm = time.perf_counter() - t0
    d1 = mc.get2DDensities([pairs[5]])[0]
    d2 = mc.get2DDensities([pairs[5]])[0]
    assert np.array_equal(d1.P, d2.P)  # the same call twice on one device: bit for bit
    # one pair against the oracle at C5's full row count (a bounded pair: no TNC in its bandwidth)
    from oracle import kde_oracle as ko

    bounded = [j for j, p in enumerate(mc.paramNames.names) if p.has_limits]
    a, b = bounded[0], bounded[1]
    sub = [names[a], names[b]]
    t0 = time.perf_counter()
    orc = ko.OracleSamples(np.ascontiguousarray(s[:, [a, b]]), names=sub, ranges={k: v for k, v in ranges.items() if k in sub})
    tr = {}
    o = orc.density_2d(0, 1, trace=tr)
    t_oracle = time.perf_counter() - t0
    d = mc.get2DDensities([(a, b)])[0]
    assert d.bandwidth_branch == tr["branch"]
    assert np.allclose(d.bandwidth, (tr["hx"], tr["hy"], tr["c"]), rtol=1e-6, atol=1e-12)
    c5_err = float(np.max(np.abs(d.P - o["P"])))
    assert c5_err < 1e-6, (sub, c5_err)
    _report("C5", dict(oracle_pair=sub, oracle_pair_max_abs_dP=c5_err, oracle_pair_cpu_s=round(t_oracle, 1), params=n, rows=N, pairs=done, generate_s=round(t_gen, 1), construct_upload_s=round(t_ctor, 1), triangle_warm_s=round(t_tri_warm, 2), densities_per_s_warm=round(done / t_tri_warm, 1),
                       margestats_200_params_s=round(t_marge, 2), triangle_s=round(t_tri, 1),
                       densities_per_s=round(done / t_tri, 1), F_classes=sorted(classes)))
