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
        assert np.allclose(a.P, b.P, rtol=0, atol=2e-3)  # TNC pairs are chaotic; the rest are bit-stable
        assert abs(a.norm_integral() - b.norm_integral()) < 1e-2 * a.norm_integral()
    p = mc.get1DDensities([0, 4, 9])
    for d in p:
        assert d.P.max() == 1.0 and d.P.shape == (1024,)


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
