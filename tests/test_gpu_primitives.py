"""GPU parity tests of the O(N) kernels, through the C ABI, against numpy / the oracle on seeded inputs."""

import os

import numpy as np
import pytest

from getdist_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from getdist_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", params=["weighted", "unit"])
def data(request, ctx):
    N = 300_001  # odd on purpose: exercises the unaligned tails
    s, w, names, ranges = synth.block_recipe(10, N, weighted=(request.param == "weighted"), stream=21)
    ctx.upload(s, w)
    return s, (w if w is not None else np.ones(N)), w is not None


def test_upload_layouts(ctx):
    r = np.random.default_rng(1)
    a = r.standard_normal((1000, 7))
    for arr in (a, np.asfortranarray(a)):
        ctx.upload(arr, None)
        st = ctx.col_stats()
        assert np.array_equal(st[:, 0], a.min(axis=0))
        assert np.array_equal(st[:, 1], a.max(axis=0))


def test_col_stats_and_cov(ctx, data):
    s, w, _ = data
    norm = w.sum()
    st = ctx.col_stats()
    assert np.array_equal(st[:, 0], s.min(axis=0))
    assert np.array_equal(st[:, 1], s.max(axis=0))
    means = w.dot(s) / norm
    assert np.allclose(st[:, 2], means, rtol=1e-12, atol=1e-14)
    var = np.array([w.dot((s[:, i] - means[i]) ** 2) / norm for i in range(s.shape[1])])
    assert np.allclose(st[:, 3], var, rtol=1e-12)
    m, cov, nrm = ctx.cov()
    d = s - means
    ref = (d * w[:, None]).T @ d / norm
    assert abs(nrm - norm) <= 1e-12 * norm
    assert np.allclose(cov, ref, rtol=1e-11, atol=1e-13)
    assert np.array_equal(cov, cov.T)
    # row sub-range and column subset (chain slices for Gelman-Rubin)
    lo, hi = 1001, 200_000
    m2, cov2, n2 = ctx.cov(cols=[3, 7, 1], lo=lo, hi=hi)
    ws, ss = w[lo:hi], s[lo:hi][:, [3, 7, 1]]
    mm = ws.dot(ss) / ws.sum()
    assert np.allclose(m2, mm, rtol=1e-12, atol=1e-14)
    dd = ss - mm
    assert np.allclose(cov2, (dd * ws[:, None]).T @ dd / ws.sum(), rtol=1e-11, atol=1e-13)
    ws_ = ctx.weight_stats(thresh=2.0)
    assert abs(ws_["norm"] - norm) <= 1e-12 * norm
    assert ws_["max_w"] == w.max()
    assert abs(ws_["sum_w2"] - w.dot(w)) <= 1e-12 * w.dot(w)
    assert ws_["n_above"] == np.sum(w > 2.0)


def test_quantiles_match_argsort_semantics(ctx, data):
    s, w, weighted = data
    fracs = np.array([0.001, 0.999] + list(np.linspace(0.1, 0.9, 9)) + [1.0, 2.0, 0.0])
    norm = np.sum(w)
    cols = [0, 4, 9]
    targets = np.tile(norm * fracs, (len(cols), 1))
    got = ctx.quantiles(cols, targets)
    for ci, c in enumerate(cols):
        idx = s[:, c].argsort()
        cum = np.cumsum(w[idx])
        ix = np.searchsorted(cum, targets[ci])
        want = s[:, c][idx[np.minimum(ix, len(idx) - 1)]]
        if weighted:  # summation order may move a knife-edge pick by one sample
            pos_got = np.searchsorted(s[:, c][idx], got[ci])
            pos_want = np.searchsorted(s[:, c][idx], want)
            assert np.all(np.abs(pos_got - pos_want) <= 1)
            assert np.mean(got[ci] == want) > 0.8
        else:
            assert np.array_equal(got[ci], want)


def test_bin_indices_bit_exact(ctx, data):
    s, w, _ = data
    for c, F in ((0, 1024), (4, 256), (7, 960)):
        x = s[:, c]
        binmin = x.min() - 0.1 * (x.max() - x.min())
        width = (x.max() + 0.1 * (x.max() - x.min()) - binmin) / (F - 1)
        idx, bad = ctx.bin_indices(c, binmin, width, F, round_half=True)
        assert bad == 0
        assert np.array_equal(idx, ((x - binmin) / width + 0.5).astype(int))
        idx, bad = ctx.bin_indices(c, binmin, width, F, round_half=False)
        assert np.array_equal(idx, ((x - binmin) / width).astype(int))


def test_hist1d(ctx, data):
    s, w, weighted = data
    cols = [0, 3, 9]
    F = 1024
    bm, wd = [], []
    for c in cols:
        x = s[:, c]
        bm.append(x.min())
        wd.append((x.max() - x.min()) / (F - 1))
    h = ctx.hist1d(cols, bm, wd, F)
    for k, c in enumerate(cols):
        ix = ((s[:, c] - bm[k]) / wd[k] + 0.5).astype(int)
        ref = np.bincount(ix, weights=w, minlength=F)
        if weighted:
            assert np.allclose(h[k], ref, rtol=1e-12, atol=1e-12)
        else:
            assert np.array_equal(h[k], ref)


@pytest.mark.parametrize("F", [256, 384, 64])
def test_hist2d_direct_and_prebinned(ctx, data, F):
    s, w, weighted = data
    pairs = [(0, 1), (5, 6), (2, 9), (9, 2)]
    bx, wx, by, wy = [], [], [], []

    def edges(c):
        x = s[:, c]
        b0 = x.min() - 0.05 * (x.max() - x.min())
        return b0, (x.max() - b0) / (F - 1)

    for a, b in pairs:
        e = edges(a), edges(b)
        bx.append(e[0][0]), wx.append(e[0][1]), by.append(e[1][0]), wy.append(e[1][1])
    d = ctx.hist2d([p[0] for p in pairs], [p[1] for p in pairs], bx, wx, by, wy, F)
    H = d.to_host((len(pairs), F, F))
    pre = {c: ctx.prebin(c, *edges(c), F) for c in {p for pr in pairs for p in pr}}
    d2 = ctx.hist2d_prebinned([pre[a] for a, b in pairs], [pre[b] for a, b in pairs], F)
    H2 = d2.to_host((len(pairs), F, F))
    for k, (a, b) in enumerate(pairs):
        ixs = ((s[:, a] - bx[k]) / wx[k] + 0.5).astype(int)
        iys = ((s[:, b] - by[k]) / wy[k] + 0.5).astype(int)
        ref = np.bincount(ixs + iys * F, weights=w, minlength=F * F).reshape(F, F)
        if weighted:
            assert np.allclose(H[k], ref, rtol=1e-12, atol=1e-12)
            assert np.allclose(H2[k], ref, rtol=1e-12, atol=1e-12)
        else:
            assert np.array_equal(H[k], ref)
            assert np.array_equal(H2[k], ref)
    # single-pair launch takes the chunked (atomic-merge) path
    d1 = ctx.hist2d([0], [1], bx[:1], wx[:1], by[:1], wy[:1], F)
    H1 = d1.to_host((1, F, F))
    assert np.allclose(H1[0], H[0], rtol=1e-12, atol=1e-12)


def test_sheared_hist_and_minmax(ctx, data):
    s, w, _ = data
    F = 256
    r0, r1 = -0.37, 1.21
    mm = ctx.minmax_affine([5], [6], [r0], [r1])
    p2 = r0 * s[:, 5] + r1 * s[:, 6]
    assert mm[0, 0] == p2.min() and mm[0, 1] == p2.max()
    from oracle.kde_oracle import trunc_bin_samples

    b1, R1 = trunc_bin_samples(s[:, 5], nbins=F)
    b2, R2 = trunc_bin_samples(p2, nbins=F)
    x = s[:, 5]
    xmin = x.min() - (x.max() - x.min()) * 0.1
    ymin = p2.min() - (p2.max() - p2.min()) * 0.1
    d = ctx.hist2d_sheared([5], [6], [r0], [r1], [xmin], [R1 / (F - 1)], [ymin], [R2 / (F - 1)], F)
    H = d.to_host((F, F))
    ref = np.bincount(b1 + b2 * F, weights=w, minlength=F * F).reshape(F, F)
    assert np.allclose(H, ref, rtol=1e-12, atol=1e-12)


def test_minmax_affine_groups_of_pairs_over_shared_columns(ctx, data):
    """Batches of >= 4 sheared pairs are evaluated in groups that read each shared column once: every pair's min / max
    must still be the exact min / max of r0*x_i + r1*x_j (same fp64 operations, no contraction), whatever the grouping."""
    s, w, _ = data
    n = s.shape[1]
    rng = np.random.default_rng(11)
    # correlated-block pattern (many pairs over few columns), a long random tail (forces several groups), a repeated pair
    pairs = [(i, j) for i in range(min(n, 5)) for j in range(i)] + [tuple(rng.choice(n, 2, replace=False)) for _ in range(37)]
    pairs += [pairs[0], (1, 1)]
    r = rng.normal(size=(len(pairs), 2))
    mm = ctx.minmax_affine([p[0] for p in pairs], [p[1] for p in pairs], r[:, 0], r[:, 1])
    for k, (i, j) in enumerate(pairs):
        p2 = r[k, 0] * s[:, i] + r[k, 1] * s[:, j]
        assert mm[k, 0] == p2.min() and mm[k, 1] == p2.max(), (k, i, j)


def test_lag_sums(ctx, data):
    s, w, _ = data
    x = s[:, 2]
    mean = w.dot(x) / w.sum()
    d = (x - mean) * w
    got = ctx.autocov_lags(2, mean, 0, 40)
    ref = np.array([np.dot(d[:len(d) - k], d[k:]) for k in range(40)])
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-9 * abs(ref[0]))
    got = ctx.autocov_lags(2, mean, 37, 5)
    assert np.allclose(got, [np.dot(d[:len(d) - k], d[k:]) for k in range(37, 42)], rtol=1e-9, atol=1e-9 * abs(ref[0]))
    lags = np.array([1, 2, 7, len(x) // 2, len(x) // 2 + 4])
    c = 1.0 / (4 * 0.3**2)
    got = ctx.kde_lag_sums(2, c, lags)
    ref = [np.dot(np.exp(-((x[:-k] - x[k:]) ** 2) * c) * w[:-k], w[k:]) for k in lags]
    assert np.allclose(got, ref, rtol=1e-12)


@pytest.mark.parametrize("N", [4096, 4097, 50_000, 50_001, 131_075])
@pytest.mark.parametrize("weighted", [False, True])
def test_neff_lag_set_reads_every_sample_once(ctx, N, weighted):
    """The lag set of the N_eff estimate (mcsamples.py:1019 ff.: five lags from N//2, one or two short ones) takes a
    kernel that walks only the first half and gives row i + N//2 its short-lag terms from the same loads: same sums as
    the general kernel and as numpy, for even and odd N (odd: the halves overlap in one row), in the caller's order."""
    r = np.random.default_rng(N + weighted)
    x = np.cumsum(r.standard_normal((N, 3)), axis=0) * 0.05 + r.standard_normal((N, 3))
    w = r.random(N) + 0.25 if weighted else None
    ctx.upload(x, w)
    ww = w if weighted else np.ones(N)
    c = np.array([1.0 / (4 * 0.3**2), 0.8, 2.5])
    for lags in ([N // 2 + d for d in range(5)] + [1, 2], [N // 2 + d for d in range(5)] + [1], [1, 2] + [N // 2 + d for d in (4, 0, 2, 1, 3)]):
        lags = np.array(lags, dtype=np.int64)
        ref = np.array([[np.dot(np.exp(-((x[:-k, j] - x[k:, j]) ** 2) * c[j]) * ww[:-k], ww[k:]) for k in lags] for j in range(3)])
        got = ctx.kde_lag_sums_batch([0, 1, 2], c, lags)
        assert np.allclose(got, ref, rtol=1e-12), (lags, np.abs(got / ref - 1).max())
        os.environ["GDHIP_KDE_LAG_UNFOLDED"] = "1"
        try:
            general = ctx.kde_lag_sums_batch([0, 1, 2], c, lags)
        finally:
            del os.environ["GDHIP_KDE_LAG_UNFOLDED"]
        assert np.allclose(got, general, rtol=1e-12)
        assert not np.array_equal(got, general)  # the other order of summation: the folded kernel did run


def test_hist2d_u16_counters_overflow_is_detected_and_redone(ctx):
    """Batched unit-weight launches use 16-bit packed LDS counters; a bin with > 65535 samples must still be exact."""
    r = np.random.default_rng(5)
    N = 400_003
    cols = [np.full(N, 0.5), np.where(r.random(N) < 0.7, 0.25, r.random(N))]  # one bin gets all / 70 % of the samples
    cols += [r.random(N) for _ in range(5)]
    s = np.column_stack(cols)
    ctx.upload(s, None)
    F = 256
    pre = [ctx.prebin(c, -0.001, 1.002 / (F - 1), F) for c in range(s.shape[1])]
    pairs = [(a, b) for a in range(7) for b in range(7) if a != b] * 7  # 294 pairs: enough blocks for the u16 path
    H = ctx.hist2d_prebinned([pre[a] for a, b in pairs], [pre[b] for a, b in pairs], F).to_host((len(pairs), F, F))
    for k, (a, b) in enumerate(pairs):
        ix = ((s[:, a] + 0.001) / (1.002 / (F - 1)) + 0.5).astype(int)
        iy = ((s[:, b] + 0.001) / (1.002 / (F - 1)) + 0.5).astype(int)
        ref = np.bincount(ix + iy * F, minlength=F * F).reshape(F, F)
        assert np.array_equal(H[k], ref), (a, b, H[k].max(), ref.max())
    assert max(H[k].max() for k in range(len(pairs))) > 65535


def test_integer_weights_take_exact_u32_path(ctx):
    """MCMC multiplicities: integral weights use u32 LDS counters (2 stripes) and must be bit-exact."""
    r = np.random.default_rng(7)
    N = 250_007
    s = r.standard_normal((N, 4))
    w = r.integers(0, 9, N).astype(float)
    ctx.upload(s, w)
    F = 256
    e = [(s[:, c].min() - 0.01, (s[:, c].max() - s[:, c].min() + 0.02) / (F - 1)) for c in range(4)]
    pairs = [(0, 1), (2, 3), (1, 3)]
    Hd = ctx.hist2d([a for a, b in pairs], [b for a, b in pairs], [e[a][0] for a, b in pairs], [e[a][1] for a, b in pairs],
                    [e[b][0] for a, b in pairs], [e[b][1] for a, b in pairs], F).to_host((3, F, F))
    pre = [ctx.prebin(c, e[c][0], e[c][1], F) for c in range(4)]
    Hp = ctx.hist2d_prebinned([pre[a] for a, b in pairs], [pre[b] for a, b in pairs], F).to_host((3, F, F))
    for k, (a, b) in enumerate(pairs):
        ix = ((s[:, a] - e[a][0]) / e[a][1] + 0.5).astype(int)
        iy = ((s[:, b] - e[b][0]) / e[b][1] + 0.5).astype(int)
        ref = np.bincount(ix + iy * F, weights=w, minlength=F * F).reshape(F, F)
        assert np.array_equal(Hd[k], ref) and np.array_equal(Hp[k], ref)
    # a non-integral weight anywhere switches back to fp64 accumulation
    w2 = w.copy()
    w2[12345] = 0.5
    ctx.upload(s, w2)
    H2 = ctx.hist2d([0], [1], [e[0][0]], [e[0][1]], [e[1][0]], [e[1][1]], F).to_host((F, F))
    ix = ((s[:, 0] - e[0][0]) / e[0][1] + 0.5).astype(int)
    iy = ((s[:, 1] - e[1][0]) / e[1][1] + 0.5).astype(int)
    assert np.allclose(H2, np.bincount(ix + iy * F, weights=w2, minlength=F * F).reshape(F, F), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("wmax", [6, 200, 300])
def test_byte_multiplicities_in_the_packed_binning(ctx, wmax):
    """
    Integer weights <= 255 ride through the batched 16-bit packed binning as one byte per sample (wmax 6, 200: the
    second overflows counters and must be detected against the accepted WEIGHT and redone); anything larger (300) keeps
    the u32 path.  All exact.
    """
    r = np.random.default_rng(13)
    N = 300_011
    cols = [np.where(r.random(N) < 0.5, 0.25, r.random(N))] + [r.random(N) for _ in range(6)]
    s = np.column_stack(cols)
    w = r.integers(0, wmax + 1, N).astype(float)
    ctx.upload(s, w)
    F = 256
    pre = [ctx.prebin(c, -0.001, 1.002 / (F - 1), F) for c in range(s.shape[1])]
    pairs = [(a, b) for a in range(7) for b in range(7) if a != b] * 7  # 294 pairs: the batched path
    H = ctx.hist2d_prebinned([pre[a] for a, b in pairs], [pre[b] for a, b in pairs], F).to_host((len(pairs), F, F))
    idx = [((s[:, c] + 0.001) / (1.002 / (F - 1)) + 0.5).astype(int) for c in range(s.shape[1])]
    for k, (a, b) in enumerate(pairs[:42]):
        ref = np.bincount(idx[a] + idx[b] * F, weights=w, minlength=F * F).reshape(F, F)
        assert np.array_equal(H[k], ref), (wmax, a, b, H[k].max(), ref.max())
    assert np.array_equal(H[:42], H[42:84])
    if wmax == 200:
        assert H.max() > 65535  # a counter did wrap and was redone


@pytest.mark.parametrize("weights", ["unit", "integer"])
def test_quantiles_with_ties_take_the_radix_fallback(weights):
    """
    Heavily tied columns overflow the collect list of the early-finish quantile select (k_qsel_collect, 4096 rows per
    bucket) and must fall back to the plain radix passes; short-bucket and tied columns share one call.  Unit and
    integer weights make the reference's pick unambiguous, so the comparison is exact.
    """
    from getdist_amd._lib import Context

    rng = np.random.default_rng(77)
    N = 250_003
    tied = rng.choice([-1.5, 0.25, 0.2500000001, 7.0], size=N, p=[0.2, 0.3, 0.3, 0.2])
    cont = rng.standard_normal(N)
    const = np.full(N, 3.25)
    few = np.round(rng.standard_normal(N), 1)  # ~80 distinct values
    s = np.column_stack([cont, tied, const, few, np.abs(cont)])
    w = None if weights == "unit" else rng.integers(1, 5, N).astype(float)
    c = Context(0)
    try:
        c.upload(s, w)
        wv = np.ones(N) if w is None else w
        fracs = np.array([0.0, 0.001, 0.2, 0.25, 0.5, 0.5000001, 0.8, 0.999, 1.0, 1.5])
        cols = [0, 1, 2, 3, 4]
        for lo, hi in ((0, N), (1001, 200_000)):
            norm = wv[lo:hi].sum()
            got = c.quantiles(cols, np.tile(norm * fracs, (len(cols), 1)), lo=lo, hi=hi)
            for ci, col in enumerate(cols):
                x = s[lo:hi, col]
                idx = x.argsort(kind="stable")
                cum = np.cumsum(wv[lo:hi][idx])
                want = x[idx[np.minimum(np.searchsorted(cum, norm * fracs), len(idx) - 1)]]
                assert np.array_equal(got[ci], want), (weights, col, lo, got[ci], want)
    finally:
        c.close()


@pytest.mark.parametrize("weights", ["unit", "integer", "real", "wide"])
def test_linear_bucket_quantiles_equal_the_radix_path(weights, monkeypatch):
    """gd_quantiles_mm (one linear-bucket counting pass + collect, given the columns' min / max) against the radix path
    and against argsort semantics: continuous, bounded, heavily tied (list overflow -> radix fallback inside the call),
    constant (degenerate range -> radix) columns in ONE call, full range and a sub-range with the full range's extrema,
    targets at 0, beyond the total weight and on the extreme rows."""
    from getdist_amd._lib import Context

    rng = np.random.default_rng(5)
    N = 2_000_003
    cont = rng.standard_normal(N)
    s = np.column_stack([cont, np.abs(cont), rng.uniform(-3, 9, N), np.round(cont, 1), np.full(N, 3.25),
                         np.where(rng.random(N) < 0.5, cont * 1e-9, cont * 1e6)])
    w = None if weights == "unit" else (rng.integers(1, 5, N).astype(float) if weights == "integer" else rng.exponential(1.0, N))
    if weights == "wide":  # importance weights over nine decades: the fixed-point bucket sums keep 2^-62 of the total
        w = w * 10.0 ** rng.uniform(-6, 3, N)
    wv = np.ones(N) if w is None else w
    c = Context(0)
    try:
        c.upload(s, w)
        cols = list(range(s.shape[1]))
        mm = np.stack([s.min(axis=0), s.max(axis=0)], axis=1)
        fracs = np.array([0.0, 1e-7, 0.001, 0.1, 0.25, 0.5, 0.5000001, 0.9, 0.999, 1.0 - 1e-9, 1.0, 1.5])
        for lo, hi in ((0, N), (12_345, 1_500_001)):
            norm = wv[lo:hi].sum()
            targets = np.tile(norm * fracs, (len(cols), 1))
            lin = c.quantiles(cols, targets, lo=lo, hi=hi, minmax=mm)
            # integer bucket sums: the same picks in every run, for every kind of weight
            assert np.array_equal(lin, c.quantiles(cols, targets, lo=lo, hi=hi, minmax=mm))
            monkeypatch.setenv("GDHIP_QSEL_RADIX", "1")
            rad = c.quantiles(cols, targets, lo=lo, hi=hi, minmax=mm)
            monkeypatch.delenv("GDHIP_QSEL_RADIX")
            plain = c.quantiles(cols, targets, lo=lo, hi=hi)
            assert np.array_equal(rad, plain)
            for ci, col in enumerate(cols):
                x = s[lo:hi, col]
                idx = x.argsort(kind="stable")
                xs = x[idx]
                cum = np.cumsum(wv[lo:hi][idx])
                want = xs[np.minimum(np.searchsorted(cum, norm * fracs), len(idx) - 1)]
                if weights in ("real", "wide"):  # the weight below a bucket is summed in another order: a knife-edge target may
                    for got in (lin[ci], rad[ci]):  # pick the neighbouring sample; both paths within one row of numpy
                        assert np.all(np.abs(np.searchsorted(xs, got) - np.searchsorted(xs, want)) <= 1), (col, lo)
                        assert np.mean(got == want) >= 0.75, (col, lo, got, want)
                else:
                    assert np.array_equal(lin[ci], want), (weights, col, lo, lin[ci], want)
                    assert np.array_equal(rad[ci], want), (weights, col, lo)
    finally:
        c.close()


@pytest.mark.parametrize("weights", ["unit", "integer"])
def test_collect_stage_overflow_falls_through_to_the_lists(weights):
    """The collect pass stages a block's hits in LDS (2048 rows) and appends them with one atomic per (block, bucket).
    Many columns in one call leave 8 blocks per column; a lattice column whose live buckets hold ~3300 rows each then
    puts ~4500 hits into one block: the rows beyond the stage go straight to the lists and the picks stay exact."""
    from getdist_amd._lib import Context

    rng = np.random.default_rng(31)
    N = 1_000_003
    s = np.column_stack([rng.integers(0, 300, N) / 7.0, rng.standard_normal(N)])
    w = None if weights == "unit" else rng.integers(1, 4, N).astype(float)
    wv = np.ones(N) if w is None else w
    c = Context(0)
    try:
        c.upload(s, w)
        cols = [0] * 255 + [1]
        mm = np.stack([s.min(axis=0), s.max(axis=0)], axis=1)[cols]
        fracs = np.array([0.003, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.997])
        for lo, hi in ((0, N), (777, 900_001)):  # whole columns (bucket columns) and a row range (fp64 collect)
            norm = wv[lo:hi].sum()
            got = c.quantiles(cols, np.tile(norm * fracs, (len(cols), 1)), lo=lo, hi=hi, minmax=mm)
            for ci in (0, 100, 254, 255):
                x = s[lo:hi, cols[ci]]
                idx = x.argsort(kind="stable")
                cum = np.cumsum(wv[lo:hi][idx])
                want = x[idx[np.minimum(np.searchsorted(cum, norm * fracs), len(idx) - 1)]]
                assert np.array_equal(got[ci], want), (weights, ci, lo, got[ci], want)
    finally:
        c.close()


@pytest.mark.parametrize("plain", [False, True])
def test_page_locked_blocks_small_and_large(plain, monkeypatch):
    """gd_host_alloc: blocks of 32 MB and more are anonymous memory on transparent huge pages registered with the runtime
    (round 6), smaller ones -- and everything with GDHIP_HOST_ALLOC_PLAIN=1 -- come from hipHostMalloc; both kinds take
    result copies and are released with the context."""
    from getdist_amd._lib import Context

    if plain:
        monkeypatch.setenv("GDHIP_HOST_ALLOC_PLAIN", "1")
    c = Context(0)
    try:
        r = np.random.default_rng(3)
        for n in (1000, 6_000_000):  # 8 KB (a 1-MB block), 48 MB
            x = r.standard_normal(n)
            d = c.alloc(n * 8)
            d.from_host(x)
            back = d.to_host((n,), pinned=True)
            assert np.array_equal(back, x)
            again = c.pinned_array((n,), np.float64)  # `back` is alive: a second block of the size
            assert again.ctypes.data != back.ctypes.data
            again[:] = 1.0
            d.free()
        del back, again  # nobody views the blocks any more: close() hands them back (hipHostUnregister + munmap / hipHostFree)
    finally:
        c.close()


def test_contour_levels_batch(ctx):
    """gd_contour_levels against the oracle's restatement of densities.py:19-56 on smooth, flat-topped and tied grids."""
    from oracle import kde_oracle as ko

    rng = np.random.default_rng(3)
    F = 96
    y, x = np.mgrid[0:F, 0:F] / (F - 1.0)
    grids = [np.exp(-((x - 0.4) ** 2 / 0.02 + (y - 0.6) ** 2 / 0.05)),
             np.exp(-((x - 0.5) ** 2 + (y - 0.5) ** 2) / 0.5),  # mass on the edges: half-edge weights matter
             np.minimum(1.0, 3 * np.exp(-((x - 0.5) ** 2 + (y - 0.5) ** 2) / 0.03)),  # flat top: thousands of exact ties at 1
             rng.random((F, F)) ** 3,
             np.round(rng.random((F, F)), 2)]  # ~100 distinct values: ties at every level
    grids[0][grids[0] < 1e-12] = 0.0
    P = np.array([g / g.max() for g in grids])
    contours = (0.68, 0.95, 0.99)
    d = ctx.alloc(P.nbytes)
    d.from_host(P)
    got, status = ctx.contour_levels(d, len(P), F, contours)
    d.free()
    for b in range(len(P)):
        want = ko.contour_levels(P[b], contours)
        if status[b] == 0:
            assert np.allclose(got[b], want, rtol=1e-9, atol=1e-12), (b, got[b], want)
        else:
            assert status[b] == -5 and b in (2, 4)  # tie list overflow: the host path takes that grid
    assert status[0] == 0 and status[1] == 0 and status[3] == 0


@pytest.mark.parametrize("F", [1024, 257, 4096, 64])
def test_limits1d_batch_matches_the_reference_algorithm(ctx, F):
    """gd_limits1d (device spline refinement + radix select + crossing search) against the oracle's restatement of
    Density1D.initLimitGrids/getLimits (densities.py:186-248: FITPACK spline, np.sort, np.cumsum) on shapes with two
    modes, a hard cut-off, a plateau of exact ties and noise; flags exact, positions to 1e-9 of the grid range."""
    from oracle import kde_oracle as ko

    r = np.random.default_rng(F)
    x = np.linspace(-3.1, 4.2, F)
    shapes = []
    for k in range(12):
        P = np.exp(-0.5 * ((x - 0.3 * k / 4) / (0.4 + 0.05 * k)) ** 2)
        if k % 2:
            P += 0.4 * np.exp(-0.5 * ((x - 2.5) / 0.3) ** 2)
        if k % 3 == 0:
            P = P * (x > -1.0)          # hard lower edge: one-tail limits
        if k % 4 == 1:
            P = np.minimum(P, 0.8)      # flat top: exactly tied values
        if k % 5 == 0:
            P = P + 0.01 * r.random(F)
        if k == 7:
            P = np.exp(-0.5 * (x / 5.0) ** 2)  # wider than the grid: both ends open
        shapes.append(P / P.max())
    P = np.array(shapes)
    contours = [0.68, 0.95, 0.99]
    x0 = np.full(len(P), x[0])
    sp = np.full(len(P), x[1] - x[0])
    for factor in (0, 3):
        got, status = ctx.limits1d(P, x0, sp, contours, factor)
        assert np.all(status == 0)
        for b in range(len(P)):
            want = ko.density_limits_1d(x, P[b], contours, factor or None)
            assert np.array_equal(got[b][:, 2:], want[:, 2:]), (F, b, got[b], want)
            assert np.max(np.abs(got[b][:, :2] - want[:, :2])) < 1e-9 * (x[-1] - x[0]), (F, b, got[b], want)


@pytest.mark.parametrize("F", [1024, 256])
def test_isj1d_device_solver_matches_scipy_path(ctx, F):
    """gd_isj1d (DCT + MINPACK hybrd n=1 + brentq re-check inside one kernel) against the oracle, which runs scipy's
    fsolve / brentq on the same histograms: the iteration paths coincide (tests/test_native_solvers.py pins the port
    bit for bit on the CPU), so the stopping points agree to the rounding of the functional (device exp / summation
    order), also for the flat shapes that leave through MINPACK's slow-progress exits and for failures (None)."""
    from oracle import kde_oracle as ko
    from oracle.fixtures import histogram_shape_zoo

    cases = list(histogram_shape_zoo(72, seed=23, F=F))
    hist = np.array([c[1] for c in cases])
    neff = np.array([c[2] for c in cases])
    h, status = ctx.isj1d(hist, neff)
    n_none = n_recheck = 0
    worst = worst_flat = 0.0
    for b, (kind, hb, nb) in enumerate(cases):
        want = ko.isj_bandwidth_binned(hb, nb)
        if want is None:
            n_none += 1
            assert status[b] != 0, (b, kind)
            continue
        assert status[b] == 0, (b, kind)
        n_recheck += want < 0.019 * nb ** (-0.2) * 1.0000001
        err = abs(h[b] - want) / abs(want)
        if kind in (3, 4):  # flat shapes: the iteration wanders through h <= 0 and leaves by a slow-progress exit; the
            worst_flat = max(worst_flat, err)  # stopping point is more sensitive to the rounding of the functional
            assert err <= 1e-4, (b, kind, h[b], want)
        else:
            worst = max(worst, err)
            assert err <= 1e-9, (b, kind, h[b], want)
    print("isj1d worst relative deviation %.2e (flat shapes %.2e), %d failures, %d near the re-check threshold"
          % (worst, worst_flat, n_none, n_recheck))


def test_byte_index_binning_matches_numpy_and_detects_wraps():
    """gd_prebin8_batch + gd_hist2d_prebinned8 (byte indices, F = 256, unit weights, packed 16-bit counters addressed
    through v_perm_b32): bit-exact histograms against numpy's bincount on the reference's index expression, the
    out-of-range report, and the exact wrap detection (one bin receiving more than 65535 samples)."""
    from getdist_amd._lib import Context, GdhipError

    N, F = 300_007, 256
    r = np.random.default_rng(5)
    s = np.column_stack([r.standard_normal(N), r.standard_normal(N) * 2 + 1, r.exponential(1.0, N), r.uniform(-1, 1, N)])
    c = Context(0)
    c.upload(np.asfortranarray(s), None)
    lo, hi = s.min(axis=0), s.max(axis=0)
    binmin = lo - 0.05 * (hi - lo)
    width = (hi + 0.05 * (hi - lo) - binmin) / (F - 1)
    bufs = [c.alloc(N + 64) for _ in range(4)]
    bad = c.prebin8_batch([0, 1, 2, 3], binmin, width, F, bufs)
    assert np.all(bad == 0)
    ix = [((s[:, j] - binmin[j]) / width[j] + 0.5).astype(int) for j in range(4)]
    for j in range(4):
        assert np.array_equal(bufs[j].to_host((N,), dtype=np.uint8), ix[j].astype(np.uint8))
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3), (3, 0), (2, 1), (1, 1)]
    H = c.hist2d_prebinned8([bufs[a] for a, b in pairs], [bufs[b] for a, b in pairs]).to_host((len(pairs), F, F))
    for k, (a, b) in enumerate(pairs):
        want = np.bincount(ix[a] + ix[b] * F, minlength=F * F).reshape(F, F)
        assert np.array_equal(H[k], want), (a, b)
    # a narrower grid leaves samples outside: reported per column
    bad = c.prebin8_batch([0], [binmin[0] + 40 * width[0]], [width[0]], F, [bufs[0]])
    assert bad[0] == np.sum(((s[:, 0] - (binmin[0] + 40 * width[0])) / width[0] + 0.5).astype(int) < 0) > 0
    # every sample in one bin: the 16-bit counter wraps and the call says so
    t = np.zeros((70_000, 2))
    c.upload(np.asfortranarray(t), None)
    b2 = [c.alloc(70_000 + 64) for _ in range(2)]
    assert np.all(c.prebin8_batch([0, 1], [-1.0, -1.0], [2.0 / 255, 2.0 / 255], F, b2) == 0)
    with pytest.raises(GdhipError) as err:
        c.hist2d_prebinned8([b2[0]], [b2[1]])
    assert err.value.code == -5
    c.close()


def test_fused_fp64_binning_packed_counters_match_the_32_bit_kernel(monkeypatch):
    """Unit weights: gd_hist2d / gd_hist2d_sheared take the packed-16-bit, chunk-reduced kernel; with GDHIP_NO_P16 the
    32-bit stripe kernel.  Same histograms bit for bit, also when a counter wraps (redo) and for F > 256 (two stripes)."""
    from getdist_amd._lib import Context

    N = 400_003
    r = np.random.default_rng(8)
    x = r.standard_normal(N)
    y = 0.8 * x + 0.6 * r.standard_normal(N)
    z = np.zeros(N)  # all in one bin: wraps
    c = Context(0)
    c.upload(np.asfortranarray(np.column_stack([x, y, z])), None)
    for F in (256, 384, 64):
        args = ([0, 0, 2], [1, 2, 2], [-6.0] * 3, [12.0 / (F - 1)] * 3, [-6.0] * 3, [12.0 / (F - 1)] * 3, F)
        sh = ([0, 1], [1, 0], [1.0, -0.3], [-0.5, 0.7], [-6.0, -6.0], [12.0 / (F - 1)] * 2, [-8.0, -8.0], [16.0 / (F - 1)] * 2, F)
        monkeypatch.delenv("GDHIP_NO_P16", raising=False)
        a0 = c.hist2d(*args).to_host((3, F, F))
        a1 = c.hist2d_sheared(*sh).to_host((2, F, F))
        monkeypatch.setenv("GDHIP_NO_P16", "1")
        b0 = c.hist2d(*args).to_host((3, F, F))
        b1 = c.hist2d_sheared(*sh).to_host((2, F, F))
        assert np.array_equal(a0, b0) and np.array_equal(a1, b1), F
        assert a0[2].max() == N and a0.sum() == 3 * N
        ixs = ((x + 6.0) / (12.0 / (F - 1)) + 0.5).astype(int)
        iys = ((y + 6.0) / (12.0 / (F - 1)) + 0.5).astype(int)
        assert np.array_equal(a0[0], np.bincount(ixs + iys * F, minlength=F * F).reshape(F, F))
    c.close()


@pytest.mark.parametrize(
    "F,wmax",
    [(256, 16), (256, 30), (256, 40), (384, 40), (100, 9), (24, 3), (101, 5), (450, 30),  # S = 288, 320, 384 (16 x M register transforms), 480, 128, 32, 128, 512 (radix passes)
     # round 6 -- frames above 512 points on 64-lane groups: S = 576, 640, 768, 960, 1152 (row twiddles read in place, 7
     # columns per block), 1152 with 63 row tiles; windows whose table does not fit the LDS (k_win_table): S = 480, 512, 1152
     (384, 70), (500, 40), (600, 60), (768, 36), (960, 45), (1000, 20), (256, 96), (256, 116), (960, 90)])
def test_lds_convolution_route_equals_the_rocfft_route(F, wmax, monkeypatch, capfd):
    """gd_density2d: the convolutions through LDS transforms (k_rows_fwd / k_col_conv / k_rows_inv) against the same call
    through rocFFT frames (GDHIP_CONV_ROCFFT=1), on random histograms with bounded and unbounded pairs, linear boundary
    correction and the multiplicative bias correction: both are exact transforms of the same sums, so the normalised
    grids agree to rounding."""
    from getdist_amd._lib import Context

    monkeypatch.setenv("GDHIP_CONV_LOG", "1")

    r = np.random.default_rng(F + wmax)
    B = 21
    yy, xx = np.mgrid[0:F, 0:F]
    hists = np.empty((B, F, F))
    for b in range(B):
        cx, cy = r.uniform(0.2, 0.8, 2) * F
        sx, sy = r.uniform(0.05, 0.2, 2) * F
        lam = 3000.0 * np.exp(-0.5 * (((xx - cx) / sx) ** 2 + ((yy - cy) / sy) ** 2))
        hists[b] = r.poisson(lam).astype(np.float64)
    smooth = r.uniform(min(2.0, 0.5 * wmax / 2.5), wmax / 2.5, B)
    rx = smooth * r.uniform(0.6, 1.0, B)
    ry = smooth * r.uniform(0.6, 1.0, B)
    rx[0] = ry[0] = wmax / 2.5  # one pair at the widest window of the class
    corr = r.uniform(-0.7, 0.7, B)
    winw = np.maximum(1, np.rint(2.5 * np.maximum(rx, ry))).astype(np.int32)
    flags = np.array([0, 1 | 64, 2 | 64, 4 | 64, 8 | 64, 5 | 64, 10 | 64, 15 | 64, 0, 0, 3 | 64] * 2, dtype=np.int32)[:B]
    ctx = Context(0)
    try:
        ctx.upload(r.standard_normal((1000, 2)), None)
        d_hist = ctx.alloc(hists.nbytes)
        d_hist.from_host(hists)
        out = {}
        for route in ("lds", "rocfft"):
            if route == "rocfft":
                monkeypatch.setenv("GDHIP_CONV_ROCFFT", "1")
            else:
                monkeypatch.delenv("GDHIP_CONV_ROCFFT", raising=False)
            res = []
            for bco, mbc in ((1, 1), (0, 0), (1, 2)):
                d_P, status = ctx.density2d(d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc)
                assert not np.any(status)
                res.append(d_P.to_host((B, F, F)).copy())
                d_P.free()
            out[route] = res
        routes = [ln.rsplit("route=", 1)[1] for ln in capfd.readouterr().err.splitlines() if ln.startswith("gdhip conv:")]
        assert routes == ["lds"] * 3 + ["rocfft"] * 3, routes  # (the first three calls did take the LDS route)
        for a, b_ in zip(out["lds"], out["rocfft"]):
            assert np.all(a.max(axis=(1, 2)) == 1.0)
            assert float(np.max(np.abs(a - b_))) < 1e-11, float(np.max(np.abs(a - b_)))
        # the same call twice: bit-equal; and a window table read from global memory gives the spectrum of the one in LDS
        monkeypatch.delenv("GDHIP_CONV_ROCFFT", raising=False)
        d_P, _ = ctx.density2d(d_hist, B, F, rx, ry, corr, winw, flags, 1, 1)
        assert np.array_equal(d_P.to_host((B, F, F)), out["lds"][0])
        d_P.free()
        monkeypatch.setenv("GDHIP_CONV_WIN_GLOBAL", "1")
        d_P, _ = ctx.density2d(d_hist, B, F, rx, ry, corr, winw, flags, 1, 1)
        assert np.array_equal(d_P.to_host((B, F, F)), out["lds"][0])
    finally:
        ctx.close()


def test_resident_fixed_point_kernel_equals_the_streamed_one(monkeypatch):
    """gd_kopt2d at F = 256: k_kopt2d_res (matrix resident in registers + LDS, one launch for the even functionals and one
    for the odd) against k_kopt2d (matrix streamed through LDS, GDHIP_KOPT_STREAMED=1) on random histograms -- peaked,
    broad, bimodal, nearly empty, a delta spike, a flat histogram (every coefficient but the mean vanishes: NaN function
    values, so the fallback time, or the solver status without one), mixed do_corr.  Same evaluation points and formulas, another summation order: t* and the even functionals to
    1e-12, the odd ones (alternating sums over the power spectrum) to 1e-8; status words and NaN pattern equal.
    Reference: kde_bandwidth.py:146-229."""
    from getdist_amd._lib import Context

    F = 256
    r = np.random.default_rng(77)
    yy, xx = np.mgrid[0:F, 0:F]
    hists = []
    for b in range(40):
        cx, cy = r.uniform(0.25, 0.75, 2) * F
        sx, sy = r.uniform(0.02, 0.25, 2) * F
        rho = r.uniform(-0.8, 0.8)
        u, v = (xx - cx) / sx, (yy - cy) / sy
        lam = np.exp(-0.5 * (u * u - 2 * rho * u * v + v * v) / (1 - rho * rho))
        if b % 5 == 0:  # a second mode
            lam = lam + 0.6 * np.exp(-0.5 * (((xx - 0.3 * F) / (0.05 * F)) ** 2 + ((yy - 0.7 * F) / (0.08 * F)) ** 2))
        lam *= 10.0 ** r.uniform(3.0, 6.5) / lam.sum()
        hists.append(r.poisson(lam).astype(np.float64))
    spike = np.zeros((F, F))
    spike[100, 57] = 1e5
    hists += [spike, np.ones((F, F)), np.ones((F, F))]
    sparse = np.zeros((F, F))
    sparse[r.integers(0, F, 40), r.integers(0, F, 40)] = 1.0
    hists.append(sparse)
    hists = np.array(hists)
    B = len(hists)
    neff = 10.0 ** r.uniform(2.5, 6.5, B)
    do_corr = (np.arange(B) % 3 != 0).astype(np.int32)
    fallback = np.full(B, 1e-4)
    fallback[-2] = 0.0  # the second flat histogram has no fallback: its status word says so
    corr = r.uniform(-0.6, 0.6, B)
    ctx = Context(0)
    try:
        ctx.upload(r.standard_normal((1000, 2)), None)
        d_hist = ctx.alloc(hists.nbytes)
        d_hist.from_host(hists)
        out = {}
        for mode in ("resident", "streamed", "resident_again"):
            if mode == "streamed":
                monkeypatch.setenv("GDHIP_KOPT_STREAMED", "1")
            else:
                monkeypatch.delenv("GDHIP_KOPT_STREAMED", raising=False)
            out[mode] = ctx.kopt2d(d_hist, B, F, neff, do_corr, fallback, corr).copy()
        a, s = out["resident"], out["streamed"]
        assert np.array_equal(a, out["resident_again"], equal_nan=True)  # deterministic
        assert np.array_equal(a[:, 7], s[:, 7]) and np.array_equal(np.isnan(a[:, :7]), np.isnan(s[:, :7]))
        assert a[-2, 7] != 0 and a[-3, 7] == 0 and a[-3, 0] == 1e-4 and np.sum(a[:, 7] == 0) >= B - 2, a[-4:, :8]
        ok = a[:, 7] == 0

        def close(x, y, tol):  # relative, and equal where the streamed kernel says exactly zero or infinity
            with np.errstate(invalid="ignore"):
                return bool(np.all((np.abs(x - y) <= tol * np.abs(y)) | (x == y) | (np.isnan(x) & np.isnan(y))))

        assert close(a[ok, 0], s[ok, 0], 1e-12), (a[ok, 0], s[ok, 0])
        assert close(a[ok, 1:4], s[ok, 1:4], 1e-12), (a[ok, 1:4], s[ok, 1:4])
        dc = ok & (do_corr == 1)
        assert np.all(np.isnan(a[ok & (do_corr == 0), 4:7]))
        assert close(a[dc, 4], s[dc, 4], 1e-12)
        # (at t* = 0, the spike, the odd sums have no Gaussian factor: sums of +-f^10 over a flat spectrum that cancel to
        # ~1e-20 of their terms -- any order of additions gives another remainder)
        dc &= a[:, 0] > 0
        assert close(a[dc, 5:7], s[dc, 5:7], 1e-8), (a[dc, 5:7], s[dc, 5:7])
    finally:
        ctx.close()


def test_byte_index_binning_in_row_chunks_sums_partial_tables_exactly():
    """Few pairs (a rank's share of a triangle): gd_hist2d_prebinned8 cuts the rows of a pair into chunks, one block each,
    and k_p8_reduce adds the packed partial tables in 32 bits -- bit-exact against numpy, a bin may hold more than 65535
    samples as long as no single chunk's counter wraps, and a wrap inside a chunk is still reported."""
    from getdist_amd._lib import Context, GdhipError

    F = 256
    r = np.random.default_rng(12)
    N = 400_003
    s = np.column_stack([r.standard_normal(N), r.standard_normal(N), r.uniform(-1, 1, N)])
    # 90 000 samples in ONE bin of the (0, 1) grid, spread evenly over the rows: more than a 16-bit counter holds in total,
    # fewer per chunk
    hot = np.arange(0, N, 4)[:90_000]
    s[hot, 0], s[hot, 1] = 0.25, -0.5
    c = Context(0)
    c.upload(np.asfortranarray(s), None)
    lo, hi = s.min(axis=0), s.max(axis=0)
    binmin = lo - 0.05 * (hi - lo)
    width = (hi + 0.05 * (hi - lo) - binmin) / (F - 1)
    bufs = [c.alloc(N + 64) for _ in range(3)]
    assert np.all(c.prebin8_batch([0, 1, 2], binmin, width, F, bufs) == 0)
    ix = [((s[:, j] - binmin[j]) / width[j] + 0.5).astype(int) for j in range(3)]
    pairs = [(0, 1), (0, 2), (1, 2), (2, 0)]
    H = c.hist2d_prebinned8([bufs[a] for a, b in pairs], [bufs[b] for a, b in pairs]).to_host((len(pairs), F, F))
    for k, (a, b) in enumerate(pairs):
        want = np.bincount(ix[a] + ix[b] * F, minlength=F * F).reshape(F, F)
        assert np.array_equal(H[k], want), (a, b)
    assert H[0].max() >= 90_000
    # all samples of a chunk in one bin: that chunk's counter wraps, and the call says so
    t = np.zeros((300_000, 2))
    c.upload(np.asfortranarray(t), None)
    b2 = [c.alloc(300_000 + 64) for _ in range(2)]
    assert np.all(c.prebin8_batch([0, 1], [-1.0, -1.0], [2.0 / 255, 2.0 / 255], F, b2) == 0)
    with pytest.raises(GdhipError) as err:
        c.hist2d_prebinned8([b2[0]], [b2[1]])
    assert err.value.code == -5
    c.close()


def test_weighted_quantiles_without_a_usable_weight_total_take_the_radix_path(monkeypatch):
    """The fixed-point bucket sums of the linear-bucket select need the total of non-negative weights; a set with a negative
    weight (not a GetDist chain, but the entry takes any array) or auxiliary weights keeps the radix select -- same answers."""
    from getdist_amd._lib import Context

    rng = np.random.default_rng(9)
    N = 300_001
    s = rng.standard_normal((N, 3))
    w = rng.exponential(1.0, N)
    w[17] = -0.25
    c = Context(0)
    try:
        c.upload(s, w)
        mm = np.stack([s.min(axis=0), s.max(axis=0)], axis=1)
        targets = np.tile(w.sum() * np.array([0.1, 0.5, 0.9]), (3, 1))
        lin = c.quantiles([0, 1, 2], targets, minmax=mm)
        monkeypatch.setenv("GDHIP_QSEL_RADIX", "1")
        assert np.array_equal(lin, c.quantiles([0, 1, 2], targets, minmax=mm))
    finally:
        c.close()


def test_partial_column_shard_lands_at_its_columns():
    """gd_upload_shard with a block in the MIDDLE of the column range (what rank r > 0 of a multi-rank job uploads before
    gd_comm_share_columns brings the rest): the block's columns are resident at their own indices -- statistics and
    quantiles of exactly those columns equal numpy's; the extra columns behind the set are zero."""
    from getdist_amd._lib import Context

    rng = np.random.default_rng(21)
    N, n, a, b = 100_003, 7, 2, 5
    s = rng.standard_normal((N, n)) * np.arange(1, n + 1) + np.arange(n)
    w = rng.integers(1, 4, N).astype(float)
    c = Context(0)
    try:
        c.upload_shard(s[:, a:b], N, n, a, w)
        cols = list(range(a, b))
        means, cov, norm, mm = c.cov(cols, minmax=True)
        assert norm == w.sum()
        assert np.allclose(means, np.average(s[:, a:b], axis=0, weights=w), rtol=1e-13, atol=1e-13)
        assert np.array_equal(mm[:, 0], s[:, a:b].min(axis=0)) and np.array_equal(mm[:, 1], s[:, a:b].max(axis=0))
        d = s[:, a:b] - means
        assert np.allclose(cov, (d * w[:, None]).T @ d / w.sum(), rtol=1e-11, atol=1e-12)
        q = c.quantiles(cols, np.tile(w.sum() * np.array([0.25, 0.5, 0.75]), (len(cols), 1)), minmax=mm)
        for ci, j in enumerate(cols):
            idx = s[:, j].argsort(kind="stable")
            want = s[:, j][idx[np.searchsorted(np.cumsum(w[idx]), w.sum() * np.array([0.25, 0.5, 0.75]))]]
            assert np.array_equal(q[ci], want)
    finally:
        c.close()


def _bucket_test_columns(N, rng):
    """Columns that stress the bucket route: a plain Gaussian, a far-offset narrow one (|mean| >> sigma: few fp64 values per
    bucket), a reflected (hard-bounded) one piled at its minimum, heavy ties, an exponential tail, a uniform."""
    g = rng.standard_normal(N)
    tied = np.round(rng.standard_normal(N) * 3) / 3.0
    tied[0], tied[1] = tied.min() - 1.5, tied.max() + 0.25  # (keeps max > min however the ties fall)
    return np.column_stack([g, 1e6 + 1e-3 * rng.standard_normal(N), np.abs(rng.standard_normal(N)), tied,
                            rng.exponential(1.0, N), rng.uniform(-2, 5, N)])


@pytest.mark.parametrize("weighted", [False, True])
def test_bucket_columns_serve_the_select_the_probe_and_bit_exact_index_columns(monkeypatch, weighted):
    """Round 6: the counting pass of the quantile select (gd_quantiles_mm_probe over whole columns) leaves each sample's
    bucket index on the device and computes the autocovariance probe from the same read.  (a) the quantiles equal the
    unfused select's exactly; (b) the probe equals gd_autocov_lags_batch to rounding; (c) byte and u16 index columns made
    FROM THE BUCKETS (table look-up, exact fp64 route for buckets on a bin edge) are bit-equal to the fp64 kernels' and to
    numpy's expression, out-of-range reports included; (d) a new upload forgets the buckets."""
    from getdist_amd._lib import Context

    monkeypatch.setenv("GDHIP_BUCKET_ROUTE_MIN", "1")  # (the route is for many columns at once; here six must take it)
    N = 400_003
    rng = np.random.default_rng(17)
    s = _bucket_test_columns(N, rng)
    w = rng.exponential(1.0, N) if weighted else None
    wn = w if weighted else np.ones(N)
    n = s.shape[1]
    c = Context(0)
    c.upload(np.asfortranarray(s), w)
    cols = list(range(n))
    mm = np.column_stack([s.min(axis=0), s.max(axis=0)])
    means = (wn[:, None] * s).sum(axis=0) / wn.sum()
    fr = np.array([0.001, 0.999] + list(np.linspace(0.1, 0.9, 9)))
    tg = np.tile(wn.sum() * fr, (n, 1))
    # (a) + (b)
    monkeypatch.setenv("GDHIP_QLIN_UNFUSED", "1")
    q_ref = c.quantiles(cols, tg, minmax=mm)
    monkeypatch.delenv("GDHIP_QLIN_UNFUSED")
    q, probe = c.quantiles_probe(cols, tg, mm, means)
    assert np.array_equal(q, q_ref)
    assert probe is not None
    lag_ref = c.autocov_lags_batch(cols, means, 0, 8)
    d = (s - means) * wn[:, None]
    for j in range(n):
        want = np.array([np.dot(d[: N - l, j], d[l:, j]) for l in range(8)])
        scale = np.abs(want[0])
        assert np.max(np.abs(probe[j] - want)) <= 1e-11 * scale, (j, probe[j], want)
        assert np.max(np.abs(probe[j] - lag_ref[j])) <= 1e-11 * scale
    # a second run delivers the same bits (fixed reduction order)
    q2, probe2 = c.quantiles_probe(cols, tg, mm, means)
    assert np.array_equal(q2, q) and np.array_equal(probe2, probe)
    # (c) index columns from the buckets against the fp64 kernels and numpy, three grids per column: the usual padded one,
    # a narrow one that leaves samples outside, and one with an awkward width
    lo, hi = mm[:, 0], mm[:, 1]
    for F, kind in ((256, "u8"), (256, "u8-narrow"), (960, "u16"), (384, "u16-narrow")):
        narrow = kind.endswith("narrow")
        binmin = lo + (0.2 if narrow else -0.05) * (hi - lo)
        width = (hi + (-0.3 if narrow else 0.05) * (hi - lo) - binmin) / (F - 1)
        ix = [((s[:, j] - binmin[j]) / width[j] + 0.5).astype(np.int64) for j in range(n)]
        if kind.startswith("u8"):
            bufs = [c.alloc(N + 64) for _ in range(n)]
            bad = c.prebin8_batch(cols, binmin, width, F, bufs)
            got = [b.to_host((N,), dtype=np.uint8) for b in bufs]
            monkeypatch.setenv("GDHIP_NO_BUCKET_COLS", "1")
            bufs2 = [c.alloc(N + 64) for _ in range(n)]
            bad2 = c.prebin8_batch(cols, binmin, width, F, bufs2)
            monkeypatch.delenv("GDHIP_NO_BUCKET_COLS")
            assert np.array_equal(bad, bad2)
            for j in range(n):
                assert np.array_equal(got[j], bufs2[j].to_host((N,), dtype=np.uint8)), (kind, j)
                assert np.array_equal(got[j], (ix[j] & 0xff).astype(np.uint8)), (kind, j)
                assert bad[j] == np.sum((ix[j] < 0) | (ix[j] >= F)), (kind, j)
            if narrow:
                assert bad.sum() > 0
        else:
            bufs = [c.alloc(2 * N + 64) for _ in range(n)]
            c.prebin_batch(cols, binmin, width, F, bufs)
            one = c.prebin(2, binmin[2], width[2], F)  # the single-column entry takes the same route
            c.sync()
            for j in range(n):
                want = np.where((ix[j] >= 0) & (ix[j] < F), ix[j], 0xFFFF).astype(np.uint16)
                assert np.array_equal(bufs[j].to_host((N,), dtype=np.uint16), want), (kind, j)
            assert np.array_equal(one.to_host((N,), dtype=np.uint16), bufs[2].to_host((N,), dtype=np.uint16))
    # lists that overflow (heavily tied data on a bin edge; here forced by a tiny capacity): the block's rows are redone exactly
    monkeypatch.setenv("GDHIP_BUCKET_LIST_CAP", "3")
    binmin = lo - 0.05 * (hi - lo)
    width = (hi + 0.05 * (hi - lo) - binmin) / 255
    bufs = [c.alloc(N + 64) for _ in range(n)]
    assert np.all(c.prebin8_batch(cols, binmin, width, 256, bufs) == 0)
    for j in range(n):
        assert np.array_equal(bufs[j].to_host((N,), dtype=np.uint8), ((s[:, j] - binmin[j]) / width[j] + 0.5).astype(np.int64).astype(np.uint8))
    monkeypatch.delenv("GDHIP_BUCKET_LIST_CAP")
    # a sub-range select must not disturb (or use) the whole-column buckets; a later prebin still agrees
    c.quantiles(cols[:2], tg[:2] * 0.25, lo=1000, hi=N // 2, minmax=mm[:2])
    binmin = lo - 0.05 * (hi - lo)
    width = (hi + 0.05 * (hi - lo) - binmin) / 255
    bufs = [c.alloc(N + 64) for _ in range(n)]
    c.prebin8_batch(cols, binmin, width, 256, bufs)
    for j in range(n):
        assert np.array_equal(bufs[j].to_host((N,), dtype=np.uint8), ((s[:, j] - binmin[j]) / width[j] + 0.5).astype(np.int64).astype(np.uint8))
    # (d) new samples: the old buckets are gone with the old set (indices follow the NEW values without a select)
    s2 = s[::-1].copy()
    c.upload(np.asfortranarray(s2), w)
    bufs = [c.alloc(N + 64) for _ in range(n)]
    c.prebin8_batch(cols, binmin, width, 256, bufs)
    for j in range(n):
        assert np.array_equal(bufs[j].to_host((N,), dtype=np.uint8), ((s2[:, j] - binmin[j]) / width[j] + 0.5).astype(np.int64).astype(np.uint8))
    c.close()


def test_real_weight_byte_index_binning_sorted_by_stripe(monkeypatch):
    """Round 6: gd_hist2d_prebinned8 with REAL weights -- samples partitioned by 64-row stripe once per y column (streaming
    counting sort with 16-entry padded runs), fixed-point weights, one ds_add_u64 per (sample, pair) -- against numpy's
    bincount on the reference's index expression (mcsamples.py:1724-1728): 1e-12 of the largest bin, every bin within
    1e-12 relative + 2 quanta absolute; pairs that share and that do not share their y column, x = y, a ragged N, reruns
    bit-equal, a scratch budget that forces several groups of keys, and the four-pass kernel (gd_hist2d_prebinned) as a
    second witness."""
    from getdist_amd._lib import Context

    N, F = 300_011, 256
    r = np.random.default_rng(9)
    s = np.column_stack([r.standard_normal(N), r.standard_normal(N) * 2 + 1, r.exponential(1.0, N), r.uniform(-1, 1, N),
                         np.abs(r.standard_normal(N))])
    w = r.exponential(1.0, N) * 10.0 ** r.uniform(-3, 2, N)  # five decades of weight
    c = Context(0)
    c.upload(np.asfortranarray(s), w)
    n = s.shape[1]
    lo, hi = s.min(axis=0), s.max(axis=0)
    binmin = lo - 0.05 * (hi - lo)
    width = (hi + 0.05 * (hi - lo) - binmin) / (F - 1)
    bufs = [c.alloc(N + 64) for _ in range(n)]
    assert np.all(c.prebin8_batch(list(range(n)), binmin, width, F, bufs) == 0)
    ix = [((s[:, j] - binmin[j]) / width[j] + 0.5).astype(int) for j in range(n)]
    pairs = [(0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 3), (3, 0), (2, 2), (4, 1), (0, 4)]
    H = c.hist2d_prebinned8([bufs[a] for a, b in pairs], [bufs[b] for a, b in pairs]).to_host((len(pairs), F, F))
    quantum = 2.0 ** -(61 - int(np.floor(np.log2(w.sum()))))
    for k, (a, b) in enumerate(pairs):
        want = np.bincount(ix[a] + ix[b] * F, weights=w, minlength=F * F).reshape(F, F)
        cnt = np.bincount(ix[a] + ix[b] * F, minlength=F * F).reshape(F, F)
        assert np.max(np.abs(H[k] - want)) <= 1e-12 * want.max(), (a, b)
        assert np.all(np.abs(H[k] - want) <= 1e-12 * want + (cnt + 2) * quantum), (a, b)
        assert abs(H[k].sum() - w.sum()) <= 1e-12 * w.sum()
    H2 = c.hist2d_prebinned8([bufs[a] for a, b in pairs], [bufs[b] for a, b in pairs]).to_host((len(pairs), F, F))
    assert np.array_equal(H, H2)  # integer sums: the same bits in every run
    monkeypatch.setenv("GDHIP_WSORT_BYTES", str(3 * (N + 50_000) * 10))  # a key or two per group
    H3 = c.hist2d_prebinned8([bufs[a] for a, b in pairs], [bufs[b] for a, b in pairs]).to_host((len(pairs), F, F))
    monkeypatch.delenv("GDHIP_WSORT_BYTES")
    assert np.array_equal(H, H3)
    # the four-pass fp64-atomic kernel over u16 index columns: the same histograms to rounding
    b16 = [c.alloc(2 * N + 64) for _ in range(n)]
    c.prebin_batch(list(range(n)), binmin, width, F, b16)
    H4 = c.hist2d_prebinned([b16[a] for a, b in pairs], [b16[b] for a, b in pairs], F).to_host((len(pairs), F, F))
    assert np.max(np.abs(H4 - H)) <= 1e-12 * H.max()
    c.close()
