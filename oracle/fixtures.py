"""
TEST INFRASTRUCTURE ONLY.  The fixture zoo: small seeded sample sets (numpy only, reproducible on any
box) that exercise every branch of the hot path.  The shapes follow the *specs* of the reference's own
fixture zoo (getdist/tests/test_distributions.py:129-257: plain / correlated / tight Gaussians,
bi/tri-modal mixtures, hard cuts on one or both axes, flat-between-bounds) and of
getdist/tests/getdist_test.py:181-225 (periodic angle), plus weighted variants.
"""

import numpy as np

from getdist_amd import synth

KW1_VARIANTS = ({}, dict(boundary_correction_order=0, mult_bias_correction_order=0),
                dict(boundary_correction_order=2, mult_bias_correction_order=1),
                dict(boundary_correction_order=1, mult_bias_correction_order=2),
                dict(smooth_scale_1D=0.3), dict(smooth_scale_1D=2))
KW2_VARIANTS = ({}, dict(boundary_correction_order=0, mult_bias_correction_order=0), dict(fine_bins_2D=64),
                dict(smooth_scale_2D=0.3), dict(mult_bias_correction_order=2))


def _rng(k):
    return np.random.default_rng(np.random.SeedSequence([synth.BASE_SEED, 1000 + k]))


def shapes_fixture(N=20000):
    """Eight parameters forming pairs of distinct 2D shapes."""
    r = _rng(1)
    cols = []
    # (s0,s1): bimodal mixture, equal widths
    comp = r.random(N) < 0.4
    cols.append(np.where(comp, r.normal(-1.2, 0.5, N), r.normal(1.0, 0.7, N)))
    cols.append(np.where(comp, r.normal(0.8, 0.4, N), r.normal(-0.5, 0.6, N)))
    # (s2,s3): tight correlation rho=0.99
    z0, z1 = r.standard_normal(N), r.standard_normal(N)
    cols.append(z0)
    cols.append(0.99 * z0 + np.sqrt(1 - 0.99**2) * z1)
    # (s4,s5): cut correlated rho=0.95 with s5>0.3 and s4<1.2 (rejection sampling)
    out4, out5 = [], []
    need = N
    while need > 0:
        a = r.standard_normal(4 * need)
        b = 0.95 * a + np.sqrt(1 - 0.95**2) * r.standard_normal(4 * need)
        keep = (b > 0.3) & (a < 1.2)
        out4.append(a[keep][:need])
        out5.append(b[keep][:need])
        need -= len(out4[-1])
    cols.append(np.concatenate(out4))
    cols.append(np.concatenate(out5))
    # (s6,s7): flat between four cuts
    cols.append(r.uniform(-1.0, 2.0, N))
    cols.append(r.uniform(0.0, 1.0, N))
    names = ["s%d" % i for i in range(8)]
    ranges = {"s4": (None, 1.2), "s5": (0.3, None), "s6": (-1.0, 2.0), "s7": (0.0, 1.0)}
    return np.column_stack(cols), names, ranges


def periodic_fixture(N=4000):
    r = _rng(2)
    angle = r.normal(0, 1, N) % (2 * np.pi)
    radius = np.abs(r.normal(2, 0.5, N))
    return np.column_stack([angle, radius]), ["angle", "radius"], {"angle": (0, 2 * np.pi, True), "radius": (0, 5)}


def loglikes_for(samples, k=7):
    """-log(posterior) column for the mean-likelihood tests: a quadratic in the first parameters plus seeded noise."""
    s = np.asarray(samples, dtype=np.float64)
    m = min(3, s.shape[1])
    z = (s[:, :m] - s[:, :m].mean(axis=0)) / s[:, :m].std(axis=0)
    return 0.5 * np.sum(z * z, axis=1) + 0.3 * _rng(k).standard_normal(len(s)) + 11.0


MEANLIKES_CASES = (("c1_bounded", ({}, dict(mult_bias_correction_order=0), dict(mult_bias_correction_order=2))),
                   ("block10_weighted", ({},)), ("shapes", ({},)), ("periodic", (dict(fine_bins=64, fine_bins_2D=32),)))


def example_mask_function(minx, miny, stepx, stepy, mask):
    """A prior cut for the mask_function tests: the region  y > 0.6 x + 0.2  is excluded (mask set to 0 there), in the
    calling convention of mcsamples.py:1767-1770 (mask[iy, ix] <-> point (minx + ix stepx, miny + iy stepy))."""
    ny, nx = mask.shape
    x = minx + np.arange(nx) * stepx
    y = miny + np.arange(ny) * stepy
    mask[y[:, None] > 0.6 * x[None, :] + 0.2] = 0


def mcmc_chains_fixture(nchains=3, N=6000, n=5):
    """
    Integer-weight (MCMC multiplicity) chains with AR(1) correlation that differs per parameter, plus a loglikes
    column: what the Raftery-Lewis / CorrSteps tests need (mcsamples.py:1039: integer weights only).
    Returns (samples, weights, loglikes, names, chain_offsets).
    """
    r = _rng(21)
    rhos = np.linspace(0.0, 0.85, n)
    chains, ws = [], []
    for c in range(nchains):
        rows = N + 500 * c
        x = np.empty((rows, n))
        e = r.standard_normal((rows, n))
        x[0] = e[0]
        for t in range(1, rows):
            x[t] = rhos * x[t - 1] + np.sqrt(1 - rhos**2) * e[t]
        chains.append(x + 0.05 * c)
        ws.append(r.geometric(0.45, rows).astype(np.float64))
    samples = np.vstack(chains)
    weights = np.concatenate(ws)
    offsets = np.concatenate([[0], np.cumsum([len(c) for c in chains])])
    loglikes = 0.5 * np.sum(samples**2, axis=1)
    return samples, weights, loglikes, ["m%d" % i for i in range(n)], offsets


# ---- the shape zoo of the reference's own test suite (getdist/tests/test_distributions.py:129-257), from its SPECS -----
def _mixture_2d(r, N, means, shapes, weights=None, xmin=None, xmax=None, ymin=None, ymax=None):
    """N draws of a 2D Gaussian mixture; components given as (sigma_x, sigma_y, correlation) or as a 2 x 2 covariance;
    hard cuts by rejection."""
    k = len(means)
    wts = np.ones(k) / k if weights is None else np.asarray(weights, dtype=float) / np.sum(weights)
    chol = []
    for sh in shapes:
        cov = np.asarray(sh, dtype=float)
        if cov.shape != (2, 2):
            sx, sy, c = sh
            cov = np.array([[sx * sx, sx * sy * c], [sx * sy * c, sy * sy]])
        chol.append(np.linalg.cholesky(cov))
    xs, ys = [], []
    need = N
    while need > 0:
        m = 2 * need + 64
        comp = r.choice(k, size=m, p=wts)
        z = r.standard_normal((m, 2))
        pts = np.empty((m, 2))
        for q in range(k):
            sel = comp == q
            pts[sel] = z[sel] @ chol[q].T + np.asarray(means[q], dtype=float)
        keep = np.ones(m, dtype=bool)
        if xmin is not None:
            keep &= pts[:, 0] > xmin
        if xmax is not None:
            keep &= pts[:, 0] < xmax
        if ymin is not None:
            keep &= pts[:, 1] > ymin
        if ymax is not None:
            keep &= pts[:, 1] < ymax
        xs.append(pts[keep, 0][:need])
        ys.append(pts[keep, 1][:need])
        need -= len(xs[-1])
    return np.concatenate(xs), np.concatenate(ys)


def _cov2(sx, sy, c):
    return np.array([[sx * sx, sx * sy * c], [sx * sy * c, sy * sy]])


def wj_zoo_2d(N=10000):
    """
    The 23 two-dimensional shapes of Test2DDistributions (test_distributions.py:154-257) -- Gaussian, bending with a hard
    edge, hammer, skew, broad tail, rotating, tight (rho 0.99 / 0.98), cut correlated, flat between four cuts, the
    Wand & Jones bi-, tri- and quadrimodal mixtures, seven half-plane-cut Gaussians -- as ONE sample set of 46 mutually
    independent columns (x_k, y_k); the pairs of interest are (2k, 2k+1).  Returns (samples, names, ranges, pairs, labels).
    """
    r = _rng(40)
    rt = np.sqrt(0.5)
    specs = [
        ("gauss", dict(means=[[0, 0]], shapes=[(0.7, 1, 0.3)])),
        ("bending", dict(means=[[0, 0], [2, 1.8]], shapes=[(rt, 1, 0.9), (1, 1, 0.8)], weights=[0.6, 0.4], xmin=-1)),
        ("hammer", dict(means=[[0, 0], [1, 1.8]], shapes=[(rt, 1, 0.9), (0.3, 1, -0.7)], weights=[0.5, 0.5])),
        ("skew", dict(means=[[0, 0], [0, 1.2]], shapes=[_cov2(rt, 1, 0.1), _cov2(rt, 1, 0.1) / 4], weights=[0.5, 0.5])),
        ("broadtail", dict(means=[[0, 0], [0, 0.2]], shapes=[_cov2(rt, 1, 0.1), _cov2(rt, 1, 0.1) * 8], weights=[0.9, 0.1])),
        ("rotating", dict(means=[[0, 0], [0, 0.2]], shapes=[(1, 1, 0.5), (2, 2, -0.5)], weights=[0.6, 0.4])),
        ("tight", dict(means=[[0, 0], [2.5, 3.5]], shapes=[(1, 1, 0.99), (1, 1.5, 0.98)], weights=[0.6, 0.4])),
        ("cutcorr", dict(means=[[0, 0]], shapes=[(0.7, 1, 0.95)], ymin=0.3, xmax=1.2)),
        ("flat", dict(means=[[0, 0]], shapes=[(1, 2, 0)], ymin=-1, ymax=2.1, xmin=-1, xmax=0.2)),
        ("bimodal1", dict(means=[[-1, 0], [1, 0]], shapes=[(2 / 3, 2 / 3, 0)] * 2)),
        ("bimodal2", dict(means=[[-1.5, 0], [1.5, 0]], shapes=[(0.25, 1, 0)] * 2)),
        ("bimodal3", dict(means=[[-1, 1], [1, -1]], shapes=[(2 / 3, 2 / 3, 0.6)] * 2)),
        ("bimodal4", dict(means=[[1, -1], [-1, 1]], shapes=[(2 / 3, 2 / 3, 0.7), (2 / 3, 2 / 3, 0)])),
        ("trimodal1", dict(means=[[-1.2, 1.2], [1.2, -1.2], [0, 0]], shapes=[(0.6, 0.6, 0.3), (0.6, 0.6, -0.6), (0.25, 0.25, 0.2)],
                           weights=[9, 9, 2])),
        ("trimodal2", dict(means=[[-1.2, 0], [1.2, 0], [0, 0]], shapes=[(0.6, 0.6, 0.7), (0.6, 0.6, 0.7), (0.25, 0.25, -0.7)])),
        ("trimodal3", dict(means=[[-1, 0], [1, 2 * np.sqrt(3) / 3], [1, -2 * np.sqrt(3) / 3]],
                           shapes=[(0.6, 0.7, 0.6), (0.6, 0.7, 0), (0.4, 0.7, 0)], weights=[3, 3, 1])),
        ("quadrimodal", dict(means=[[-1, 1], [-1, -1], [1, -1], [1, 1]],
                             shapes=[(2 / 3, 2 / 3, 0.4), (2 / 3, 2 / 3, 0.6), (2 / 3, 2 / 3, -0.7), (2 / 3, 2 / 3, -0.5)],
                             weights=[1, 3, 1, 3])),
    ]
    for cut in (-2, -1, -0.5, 0, 1, 1.5, 2):
        specs.append(("cutx%g" % cut, dict(means=[[0, 0]], shapes=[(0.7, 1, 0.3)], xmin=cut)))
    cols, names, ranges, labels = [], [], {}, []
    for k, (label, sp) in enumerate(specs):
        x, y = _mixture_2d(r, N, **sp)
        cols += [x, y]
        nx, ny = "x%d" % k, "y%d" % k
        names += [nx, ny]
        labels.append(label)
        if sp.get("xmin") is not None or sp.get("xmax") is not None:
            ranges[nx] = (sp.get("xmin"), sp.get("xmax"))
        if sp.get("ymin") is not None or sp.get("ymax") is not None:
            ranges[ny] = (sp.get("ymin"), sp.get("ymax"))
    pairs = [(2 * k, 2 * k + 1) for k in range(len(specs))]
    return np.column_stack(cols), names, ranges, pairs, labels


def wj_zoo_1d(N=10000):
    """The 17 one-dimensional shapes of Test1DDistributions (test_distributions.py:129-151): Gaussian, skew, tailed, broad,
    flat between two cuts, flat top, two bimodal, one trimodal, six half-line-cut Gaussians -- one column each."""
    r = _rng(41)

    def mix(means, sigmas, weights=None, xmin=None, xmax=None):
        k = len(means)
        wts = np.ones(k) / k if weights is None else np.asarray(weights, dtype=float) / np.sum(weights)
        out = []
        need = N
        while need > 0:
            m = 2 * need + 64
            comp = r.choice(k, size=m, p=wts)
            v = r.standard_normal(m) * np.asarray(sigmas, dtype=float)[comp] + np.asarray(means, dtype=float)[comp]
            keep = np.ones(m, dtype=bool)
            if xmin is not None:
                keep &= v > xmin
            if xmax is not None:
                keep &= v < xmax
            out.append(v[keep][:need])
            need -= len(out[-1])
        return np.concatenate(out)

    specs = [("gauss", dict(means=[0], sigmas=[0.5])), ("skew", dict(means=[0, 1], sigmas=[1, 0.4], weights=[0.6, 0.4])),
             ("tailed", dict(means=[0, 0], sigmas=[1, 3], weights=[0.8, 0.2])),
             ("broad", dict(means=[0, 0.3], sigmas=[1, 2], weights=[0.6, 0.4])),
             ("flat", dict(means=[0], sigmas=[3], xmin=-1, xmax=2)),
             ("flattop", dict(means=[0, 1.5, 3], sigmas=[1, 1, 1], weights=[0.4, 0.2, 0.4])),
             ("bimodal1", dict(means=[0, 2], sigmas=[0.5, 0.5], weights=[0.6, 0.4])),
             ("bimodal2", dict(means=[0, 2], sigmas=[0.2, 0.5], weights=[0.5, 0.5])),
             ("trimodal", dict(means=[0, 2, 5], sigmas=[0.2, 0.7, 0.4]))]
    for cut in (-1.5, -1, -0.5, 0, 1, 1.5):
        specs.append(("cut%g" % cut, dict(means=[0], sigmas=[1], xmin=cut)))
    specs += [("gauss_b", dict(means=[1], sigmas=[2])), ("twocut", dict(means=[0], sigmas=[1], xmin=-0.7, xmax=1.1))]
    cols, names, ranges = [], [], {}
    for label, sp in specs:
        cols.append(mix(**sp))
        names.append("u_" + label)
        if sp.get("xmin") is not None or sp.get("xmax") is not None:
            ranges[names[-1]] = (sp.get("xmin"), sp.get("xmax"))
    return np.column_stack(cols), names, ranges


def ingestion_cases():
    """Array-input cases for the per-chain ingestion pipeline: (label, kwargs for MCSamples) -- shared with the tests."""
    rng = np.random.default_rng(4242)
    chains = [rng.standard_normal((n, 4)) * [1, 2, 0.5, 1] + [0, 1, -1, 3] for n in (1000, 1300, 700)]
    for c in chains:
        c[:, 2] = 0.25  # never moves in any chain: deleted, recorded as a fixed range
    chains[1][:, 3] = 3.0  # constant in chain 2 only: decided on chain 1, so it stays
    w = [rng.integers(0, 5, size=c.shape[0]).astype(float) for c in chains]  # zeros included
    w[0][:10] = 0.0
    L = [rng.random(c.shape[0]) * 10 for c in chains]
    names = ["a", "b", "c", "d"]
    cases = []
    for ign in (0, 0.3, 25):
        cases.append(("chains ignore_rows=%s" % ign, dict(samples=chains, weights=w, loglikes=L, names=names, ignore_rows=ign)))
        cases.append(("chains settings ignore_rows=%s" % ign, dict(samples=chains, weights=w, loglikes=L, names=names,
                                                                   settings={"ignore_rows": ign})))
        cases.append(("single ignore_rows=%s" % ign, dict(samples=chains[0], weights=w[0], loglikes=L[0], names=names,
                                                          ignore_rows=ign)))
    cases.append(("chains unweighted", dict(samples=chains, loglikes=L, names=names, ignore_rows=0.2)))
    return cases


def histogram_shape_zoo(n_cases, seed=5, F=1024):
    """Histograms of Gaussian, bimodal, half-Gaussian at an edge, FLAT (uniform between bounds / over the whole range:
    the cases where fsolve wanders to h <= 0 and leaves through MINPACK's slow-progress exits), exponential, mixed and
    U-shaped samples, with random sizes and effective sample numbers."""
    rng = np.random.default_rng(seed)
    for k in range(n_cases):
        kind = k % 8
        N = int(10 ** rng.uniform(3, 6))
        if kind == 0:
            s = rng.normal(0.5, 0.08, N)
        elif kind == 1:
            s = np.concatenate([rng.normal(0.3, 0.05, N // 2), rng.normal(0.7, 0.07, N - N // 2)])
        elif kind == 2:
            s = np.abs(rng.normal(0, 0.2, N)) + 0.1
        elif kind == 3:
            s = rng.uniform(0.1, 0.9, N)
        elif kind == 4:
            s = rng.uniform(0.0, 1.0, N)
        elif kind == 5:
            s = rng.exponential(0.1, N) + 0.05
        elif kind == 6:
            s = np.concatenate([rng.uniform(0.2, 0.4, N // 2), rng.normal(0.7, 0.02, N - N // 2)])
        else:
            s = rng.beta(0.5, 0.5, N)
        s = s[(s >= 0) & (s <= 1)]
        hist = np.bincount((s * (F - 1) + 0.5).astype(int), minlength=F).astype(float)
        yield kind, hist, len(s) / rng.uniform(1, 30)


def random_psi_tuples(n, seed=3, wide=False):
    """
    Random but realistic inputs of KernelOptimizer2D.get_h: the psi functionals of a correlated bivariate Gaussian of
    widths (sx, sy) in bin-range units and correlation rho, each perturbed by 20-30 %, an effective sample number and
    the sample correlation handed to the optimiser.  Yields (psi = (p02, p20, p11, p00, p13, p31), N, corr).
    """
    rng = np.random.default_rng(seed)
    made = 0
    while made < n:
        N = 10 ** (rng.uniform(1.5, 7.5) if wide else rng.uniform(2.5, 6.5))
        sx, sy = (10 ** rng.uniform(-2.3, -0.4, 2)) if wide else rng.uniform(0.02, 0.25, 2)
        rho = rng.uniform(-0.95, 0.95)
        p40 = 3 / (16 * np.pi * sx**5 * sy) * (1 + rng.normal() * 0.2)
        p04 = 3 / (16 * np.pi * sy**5 * sx) * (1 + rng.normal() * 0.2)
        p22 = 1 / (16 * np.pi * sx**3 * sy**3) * (1 + rng.normal() * 0.3) * (1 + 2 * rho**2)
        p13 = 3 * rho / (16 * np.pi * sx**2 * sy**4) * (1 + rng.normal() * 0.3)
        p31 = 3 * rho / (16 * np.pi * sx**4 * sy**2) * (1 + rng.normal() * 0.3)
        if p40 <= 0 or p04 <= 0 or p22 + np.sqrt(p40 * p04) <= 0:
            continue
        corr = float(rho * rng.uniform(0.5, 1.0)) if rng.random() < 0.85 else 0.0
        made += 1
        yield (float(p04), float(p40), float(p22), -1.0, float(p13), float(p31)), float(N), corr


def fixture_zoo():
    """Yields dicts(name, samples, weights, names, ranges, pairs, kw1, kw2)."""
    zoo = []
    s, w, names, ranges = synth.config_c1(100_000)
    zoo.append(dict(name="c1_100k", samples=s, weights=w, names=names, ranges=ranges,
                    pairs=[(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)], kw1=({},), kw2=({},)))
    s, w, names, ranges = synth.config_c1(20_000, bounded=True)
    zoo.append(dict(name="c1_bounded", samples=s, weights=w, names=names, ranges=ranges,
                    pairs=[(0, 3), (2, 3), (3, 2)], kw1=KW1_VARIANTS, kw2=KW2_VARIANTS))
    s, w, names, ranges = synth.block_recipe(10, 20_000, weighted=True, stream=11)
    zoo.append(dict(name="block10_weighted", samples=s, weights=w, names=names, ranges=ranges,
                    pairs=[(0, 1), (0, 5), (5, 6), (4, 9), (3, 4), (8, 9)], kw1=KW1_VARIANTS[:2], kw2=KW2_VARIANTS[:2]))
    s, w, names, ranges = synth.block_recipe(50, 20_000, weighted=False, stream=12)
    zoo.append(dict(name="block50", samples=s, weights=w, names=names, ranges=ranges,
                    pairs=[(15, 16), (20, 21), (25, 26), (30, 31), (35, 36), (40, 41), (38, 39), (48, 49), (39, 49),
                           (18, 19), (24, 29), (16, 19), (21, 24), (36, 39)], kw1=({},), kw2=({},)))
    s, names, ranges = shapes_fixture()
    zoo.append(dict(name="shapes", samples=s, weights=None, names=names, ranges=ranges,
                    pairs=[(0, 1), (2, 3), (4, 5), (6, 7), (0, 6), (5, 7)], kw1=KW1_VARIANTS, kw2=KW2_VARIANTS[:3]))
    r = _rng(3)
    zoo.append(dict(name="shapes_intweights", samples=s, weights=r.integers(1, 6, len(s)).astype(float), names=names,
                    ranges=ranges, pairs=[(0, 1), (4, 5)], kw1=({},), kw2=({},)))
    s, names, ranges, pairs, _ = wj_zoo_2d()
    zoo.append(dict(name="wj2d", samples=s, weights=None, names=names, ranges=ranges, pairs=pairs, kw1=({},), kw2=({},)))
    r = _rng(42)
    sel = [0, 1, 4, 5, 12, 13, 14, 15, 16, 17, 26, 27, 32, 33, 36, 37]  # gauss, hammer, tight, cutcorr, flat, trimodal1, quadrimodal, cutx-0.5
    zoo.append(dict(name="wj2d_weighted", samples=s[:, sel], weights=r.exponential(1.0, len(s)), names=[names[c] for c in sel],
                    ranges={k: v for k, v in ranges.items() if k in [names[c] for c in sel]},
                    pairs=[(2 * k, 2 * k + 1) for k in range(len(sel) // 2)], kw1=({},), kw2=({},)))
    s, names, ranges = wj_zoo_1d()
    zoo.append(dict(name="wj1d", samples=s, weights=None, names=names, ranges=ranges, pairs=[],
                    kw1=({}, dict(boundary_correction_order=2, mult_bias_correction_order=1), dict(mult_bias_correction_order=0)),
                    kw2=({},)))
    s, names, ranges = periodic_fixture()
    zoo.append(dict(name="periodic", samples=s, weights=None, names=names, ranges=ranges,
                    pairs=[(0, 1), (1, 0)], kw1=({}, dict(fine_bins=64)), kw2=(dict(fine_bins_2D=32), dict(fine_bins_2D=64))))
    return zoo
