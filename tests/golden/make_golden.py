"""
Generates tests/golden/*.npz by importing the REAL GetDist 1.7.7 from /root/reference (build container
only; the reference never travels).  Inputs are regenerated on any box from seeds by
oracle/fixtures.py + getdist_amd/synth.py, so only reference OUTPUTS are stored here.

    python tests/golden/make_golden.py
"""

import logging
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from getdist import MCSamples  # noqa: E402  (the reference)
from getdist.convolve import nearestFFTnumber  # noqa: E402

from getdist_amd import synth  # noqa: E402
from oracle.fixtures import fixture_zoo  # noqa: E402

logging.getLogger().setLevel(logging.ERROR)

PAR_ATTS = ("param_min", "param_max", "range_min", "range_max", "sigma_range", "err", "mean", "has_limits_bot",
            "has_limits_top", "N_eff_kde", "kde_h")
FULL_GRID_PAIRS = 1  # pairs per fixture whose full default-settings P grid is stored (others: strided)


def kwkey(kw):
    return ",".join("%s=%s" % (k, kw[k]) for k in sorted(kw)) or "default"


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def golden_for_fixture(name, samples, weights, names, ranges, pairs, kw1, kw2):
    out = {}
    ref = MCSamples(samples=np.ascontiguousarray(samples), weights=weights, names=names, ranges=ranges)
    out["means"] = ref.means
    out["vars"] = ref.vars
    out["cov"] = ref.fullcov
    out["corr"] = ref.getCorrelationMatrix()
    fracs = np.array([0.001, 0.999] + list(np.linspace(0.1, 0.9, 9)))
    out["quantile_fracs"] = fracs
    out["quantiles"] = np.array([ref.confidence(j, fracs) for j in range(len(names))])
    for j, nm in enumerate(names):
        for kw in kw1:
            d = ref.get1DDensityGridData(nm, **kw)
            key = "p1d/%s/%s" % (nm, kwkey(kw))
            out[key + "/P"] = d.P
            out[key + "/x0x1"] = np.array([d.x[0], d.x[-1]])
            if not kw:
                par = ref.paramNames.parWithName(nm)
                out["par/%s" % nm] = np.array([float(getattr(par, a)) for a in PAR_ATTS])
                ix, fw, bmin, bmax = ref._binSamples(ref.samples[:, j], par, ref.fine_bins)
                out["bin1d/%s/edges" % nm] = np.array([bmin, bmax, fw])
                out["bin1d/%s/ix_head" % nm] = ix[:1024].astype(np.int32)
                out["bin1d/%s/ix_crc" % nm] = crc(ix.astype(np.int32))
                out["hist1d/%s" % nm] = np.bincount(ix, weights=ref.weights, minlength=ref.fine_bins)
    # capture the bandwidth triple of every 2D call
    captured = []
    orig = ref.getAutoBandwidth2D

    def spy(*a, **k):
        r = orig(*a, **k)
        captured.append(r)
        return r

    ref.getAutoBandwidth2D = spy
    for ip, (a, b) in enumerate(pairs):
        for kw in kw2:
            captured.clear()
            d = ref.get2DDensityGridData(names[a], names[b], get_density=False, **kw)
            key = "p2d/%s/%s/%s" % (names[a], names[b], kwkey(kw))
            F = d.P.shape[0]
            out[key + "/F"] = np.int32(F)
            st = max(1, F // 64)
            out[key + "/stride"] = np.int32(st)
            if (not kw and ip < FULL_GRID_PAIRS and F <= 256) or F <= 64:
                out[key + "/P"] = d.P
            else:
                out[key + "/Pstrided"] = d.P[::st, ::st].copy()
            out[key + "/Psum"] = np.float64(np.sum(d.P))
            out[key + "/contours"] = np.asarray(d.contours, dtype=float)
            out[key + "/xy"] = np.array([d.x[0], d.x[-1], d.y[0], d.y[-1]])
            if captured:
                out[key + "/hxhyc"] = np.array(captured[0], dtype=float)
            if not kw:
                parx, pary = ref.paramNames.parWithName(names[a]), ref.paramNames.parWithName(names[b])
                ixs = ref._binSamples(ref.samples[:, a], parx, F)[0]
                iys = ref._binSamples(ref.samples[:, b], pary, F)[0]
                hist, flat = ref._make2Dhist(ixs, iys, F, F)
                out[key + "/flatix_crc"] = crc(flat.astype(np.int32))
                out[key + "/hist_blocksum"] = hist.reshape(F // st, st, F // st, st).sum(axis=(1, 3))
    return out


def golden_margestats(fx):
    ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
    ms = ref.getMargeStats()
    out = {}
    for nm in fx["names"]:
        par = ms.parWithName(nm)
        out["lims/" + nm] = np.array([[lim.lower, lim.upper, lim.twotail, lim.onetail_upper, lim.onetail_lower]
                                      for lim in par.limits], dtype=float)
        out["meanerr/" + nm] = np.array([par.mean, par.err])
    return out


def golden_split_tests(fx, test_confidence=0.95, max_split_tests=4):
    """SplitTest numbers recomputed with the reference's own confidence()/getFractionIndices (mcsamples.py:1005-1034)."""
    ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
    limits = np.array([1 - (1 - test_confidence) / 2, (1 - test_confidence) / 2])
    out = np.zeros((ref.n, max_split_tests - 1, 2))
    fracs = [ref.getFractionIndices(ref.weights, i + 2) for i in range(max_split_tests - 1)]
    for j in range(ref.n):
        confids = ref.confidence(ref.samples[:, j], limits)
        for ix, frac in enumerate(fracs):
            for f1, f2 in zip(frac[:-1], frac[1:]):
                out[j, ix, :] += (ref.confidence(ref.samples[:, j], limits, start=f1, end=f2) - confids) ** 2
            out[j, ix, :] = np.sqrt(out[j, ix, :] / (2 + ix)) / ref.sddev[j]
    return out


def golden_meanlikes(zoo):
    """Mean-likelihood grids of the reference (mcsamples.py:1556-1561,1672-1684,1829-1831,1886-1903,2004-2006)."""
    from oracle.fixtures import MEANLIKES_CASES, loglikes_for

    out = {}
    for nm, kws in MEANLIKES_CASES:
        fx = zoo[nm]
        ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"],
                        ranges=fx["ranges"], loglikes=loglikes_for(fx["samples"]))
        for kw in kws:
            kw1 = {k: v for k, v in kw.items() if k != "fine_bins_2D"}
            kw2 = {k: v for k, v in kw.items() if k != "fine_bins"}
            for shade in (False, True):
                ref.shade_likes_is_mean_loglikes = shade
                for j, name in enumerate(fx["names"][:6]):
                    out["%s/%s/1d/%d/shade%d" % (nm, kwkey(kw), j, shade)] = ref.get1DDensityGridData(
                        name, meanlikes=True, **kw1).likes
            ref.shade_likes_is_mean_loglikes = False
            for a, b in fx["pairs"][:3]:
                d = ref.get2DDensityGridData(fx["names"][a], fx["names"][b], meanlikes=True, **kw2)
                st = max(1, d.likes.shape[0] // 64)
                out["%s/%s/2d/%d_%d/stride" % (nm, kwkey(kw), a, b)] = np.int32(st)
                out["%s/%s/2d/%d_%d/likes" % (nm, kwkey(kw), a, b)] = d.likes[::st, ::st].copy()
                out["%s/%s/2d/%d_%d/sum" % (nm, kwkey(kw), a, b)] = np.float64(np.sum(d.likes))
    return out


def golden_nd_ranges(zoo):
    """ND confidence-region limits (mcsamples.py:2263-2274) and the ranges they widen with range_ND_contour = k."""
    from oracle.fixtures import loglikes_for

    out = {}
    for nm in ("block10_weighted", "shapes", "c1_bounded"):
        fx = zoo[nm]
        ll = loglikes_for(fx["samples"])
        for k in (0, 1, 2):
            ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"],
                            ranges=fx["ranges"], loglikes=ll, settings={"range_ND_contour": k})
            pars = [ref._initParamRanges(j) for j in range(len(fx["names"]))]
            out["%s/%d/range_min" % (nm, k)] = np.array([p.range_min for p in pars], dtype=float)
            out["%s/%d/range_max" % (nm, k)] = np.array([p.range_max for p in pars], dtype=float)
            if k == 0:
                out["%s/ND_limit_bot" % nm] = np.array([p.ND_limit_bot for p in pars])
                out["%s/ND_limit_top" % nm] = np.array([p.ND_limit_top for p in pars])
    return out


def golden_raftery_lewis():
    """
    Raftery-Lewis table parsed from the reference's getConvergeTests text (mcsamples.py:1039-1165) and the CorrSteps
    numbers recomputed to full precision with the reference's own thin_indices / chain diffs (:1183-1210; the text
    only carries three decimals).
    """
    from oracle.validate_against_reference import parse_raftery_lewis, reference_chain_set

    ref, samples, weights, loglikes, names, offsets = reference_chain_set()
    out = {}
    for tc in (0.95, 0.8):
        out["table/%g" % tc] = parse_raftery_lewis(ref.getConvergeTests(test_confidence=tc, what=("RafteryLewis",)))
        out["indep_thin/%g" % tc] = np.int64(ref.RL_indep_thin)
    chainlist = ref.getSeparateChains()
    for ch in chainlist:
        ch.setDiffs()
    for autocorr_thin in (20, 3):
        thin_rows = len(ref.thin_indices(autocorr_thin))
        maxoff = int(min(ref.corr_length_steps, thin_rows // (2 * len(chainlist))))
        corrs = np.zeros([maxoff, ref.n])
        for chain in chainlist:
            thin_ix = chain.thin_indices(autocorr_thin)
            thin_rows = len(thin_ix)
            maxoff = min(maxoff, thin_rows // autocorr_thin)
            for j in range(ref.n):
                diff = chain.diffs[j][thin_ix]
                for off in range(1, maxoff + 1):
                    corrs[off - 1][j] += np.dot(diff[off:], diff[:-off]) / (thin_rows - off) / ref.vars[j]
        corrs /= len(chainlist)
        out["corrsteps/%d" % autocorr_thin] = corrs[:maxoff]
    rng = np.random.default_rng(9)
    for k in range(6):  # thin indices of the reference on random multiplicities
        w = rng.geometric(0.35, 3000 + 17 * k).astype(float)
        f = [1, 2, 5, int(w.max()), int(w.max()) + 3, 40][k]
        out["thin/%d/w" % k] = w
        out["thin/%d/factor" % k] = np.int64(f)
        out["thin/%d/ix" % k] = ref.thin_indices(f, w)
    return out


def golden_mask_function(zoo):
    """get2DDensityGridData(mask_function=example_mask_function) of the reference (mcsamples.py:1907-1919,1973-1979)."""
    from oracle.fixtures import example_mask_function

    out = {}
    for nm, pairs in (("c1_bounded", [(0, 3), (2, 3)]), ("shapes", [(0, 1), (6, 7)])):
        fx = zoo[nm]
        ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
        for a, b in pairs:
            for kw in ({}, dict(mult_bias_correction_order=0), dict(boundary_correction_order=0, mult_bias_correction_order=2)):
                d = ref.get2DDensityGridData(fx["names"][a], fx["names"][b], get_density=True,
                                             mask_function=example_mask_function, **kw)
                key = "%s/%d_%d/%s" % (nm, a, b, kwkey(kw))
                out[key + "/P"] = d.P[::4, ::4].copy()
                out[key + "/Psum"] = np.float64(np.sum(d.P))
                out[key + "/mask_crc"] = crc(np.asarray(d.mask, dtype=np.uint8))
    return out


MASK_CORNER_KW_PERIODIC = ({"fine_bins_2D": 64}, dict(fine_bins_2D=64, mult_bias_correction_order=0),
                           dict(fine_bins_2D=32, boundary_correction_order=0, mult_bias_correction_order=2))
MASK_CORNER_KW_LIKES = ({}, dict(mult_bias_correction_order=0))


def golden_mask_corners(zoo):
    """mask_function on periodic parameters (either orientation) and together with meanlikes, of the reference
    (mcsamples.py:1874-1903, 1907-1987, 2004-2006).  Written to mask_function_corners.npz by --mask-corners."""
    from oracle.fixtures import example_mask_function, loglikes_for

    out = {}
    fx = zoo["periodic"]
    ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for a, b in fx["pairs"]:
        for kw in MASK_CORNER_KW_PERIODIC:
            d = ref.get2DDensityGridData(fx["names"][a], fx["names"][b], get_density=True, mask_function=example_mask_function, **kw)
            key = "periodic/%d_%d/%s" % (a, b, kwkey(kw))
            out[key + "/P"] = d.P.copy()
            out[key + "/mask_crc"] = crc(np.asarray(d.mask, dtype=np.uint8))
    fx = zoo["c1_bounded"]
    ref = MCSamples(samples=np.ascontiguousarray(fx["samples"]), weights=fx["weights"], names=fx["names"], ranges=fx["ranges"],
                    loglikes=loglikes_for(fx["samples"]))
    for a, b in ((0, 3), (2, 3)):
        for kw in MASK_CORNER_KW_LIKES:
            d = ref.get2DDensityGridData(fx["names"][a], fx["names"][b], meanlikes=True, mask_function=example_mask_function, **kw)
            key = "c1_bounded/%d_%d/%s" % (a, b, kwkey(kw))
            out[key + "/P"] = d.P[::4, ::4].copy()
            out[key + "/likes"] = d.likes[::4, ::4].copy()
            out[key + "/mask_crc"] = crc(np.asarray(d.mask, dtype=np.uint8))
    return out


def golden_convergence():
    samples, weights, names, offsets = synth.config_c4(nchains=4, N=20000, n=8)
    chains = [np.ascontiguousarray(samples[a:b]) for a, b in zip(offsets[:-1], offsets[1:])]
    ws = [weights[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    ref = MCSamples(samples=chains, weights=ws, loglikes=[np.zeros(len(w)) for w in ws], names=names)
    out = dict(gr_eigenvalues=ref.getGelmanRubinEigenvalues(), gr=np.float64(ref.getGelmanRubin()),
               means=ref.means, cov=ref.fullcov)
    chainlist = ref.getSeparateChains()
    for ch in chainlist:
        ch.setDiffs()
    between = np.zeros(ref.n)
    within = np.zeros(ref.n)
    for ch in chainlist:
        between += (ch.getMeans() - ref.means) ** 2
    between /= len(chainlist) - 1
    for j in range(ref.n):
        for ch in chainlist:
            within[j] += np.dot(ch.weights, ch.diffs[j] ** 2)
        within[j] /= ref.norm
    out["meanvar"] = np.sqrt(between / within)
    # CorrLengths block (mcsamples.py:941-962)
    maxoff = np.min([chain.weights.size // 10 for chain in chainlist])
    lens = []
    for j in range(ref.n):
        corr = np.zeros(maxoff + 1)
        for chain in chainlist:
            corr += chain.getAutocorrelation(j, maxoff, normalized=False) * chain.norm
        corr /= ref.norm * ref.vars[j]
        ix = np.argmin(corr > 0.05 * corr[0])
        lens.append(corr[0] + 2 * np.sum(corr[1:ix]))
    out["corr_lengths"] = np.array(lens)
    return out


def golden_reference_unit_tests():
    """The inputs and the asserted numbers of the reference's OWN unit tests for this path (getdist/tests/getdist_test.py):
    testFileLoadPlot (three 4000-sample text chains of `bimodal[0]`, seed 10, reloaded with ignore_rows = 0.1:
    GelmanRubin = 0.00052997 to 4 places) and testLimits (cut_correlated, 12000 samples, seed 10: limits 0.2077 /
    0.0574 to 3 places, third limit one-tailed).  Stored: the chains exactly as loadMCSamples returns them (the text
    round trip is part of the test), the sample array of testLimits, and the numbers the reference computes from them."""
    import shutil
    import tempfile

    from getdist import loadMCSamples
    from getdist.tests.test_distributions import Test2DDistributions

    out = {}
    rng = np.random.default_rng(10)
    prob = Test2DDistributions().bimodal[0]
    tmp = tempfile.mkdtemp()
    try:
        root = os.path.join(tmp, "testchain")
        for n in range(3):
            prob.MCSamples(4000, logLikes=True, random_state=rng).saveAsText(root, chain_index=n)
        raw = loadMCSamples(root, no_cache=True)  # before the burn-in: what a loader has to deliver
        out["fileload/samples"], out["fileload/loglikes"] = raw.samples, raw.loglikes
        out["fileload/weights"] = raw.weights
        out["fileload/chain_offsets"] = np.asarray(raw.chain_offsets)
        samples = loadMCSamples(root, settings={"ignore_rows": 0.1}, no_cache=True)
        samples.getConvergeTests(0.95)
        out["fileload/numrows_after_burn"] = np.int64(samples.numrows)
        out["fileload/GelmanRubin"] = np.float64(samples.GelmanRubin)
        out["fileload/asserted"] = np.float64(0.00052997)
    finally:
        shutil.rmtree(tmp)
    tl = Test2DDistributions().cut_correlated.MCSamples(12000, logLikes=False, random_state=10)
    out["limits/samples"] = tl.samples
    out["limits/names"] = np.array([p.name for p in tl.paramNames.names])
    out["limits/range_lo"] = np.array([np.nan if tl.ranges.getLower(n) is None else tl.ranges.getLower(n) for n in out["limits/names"]])
    out["limits/range_hi"] = np.array([np.nan if tl.ranges.getUpper(n) is None else tl.ranges.getUpper(n) for n in out["limits/names"]])
    lims = tl.getMargeStats().parWithName("x").limits
    out["limits/x_lower"] = np.array([lims[0].lower, lims[1].lower])
    out["limits/x_onetail_lower_2"] = np.bool_(lims[2].onetail_lower)
    out["limits/asserted"] = np.array([0.2077, 0.0574])
    return out


def golden_mutators():
    """Reference outputs after every mutator of the sample set named in SURVEY.md 8b (chains.py:941-1061,1354,
    mcsamples.py:533,2560), of getLikeStats (mcsamples.py:2216-2261,2369), of getAutocorrelation / getCorrelationLength
    with many lags, vectors and corr= (chains.py:423-466), and of the public functions of convolve.py on seeded inputs."""
    from getdist import convolve as rconv

    from make_golden_inputs import mutator_inputs

    samples, weights, loglikes, names, offsets, extra = mutator_inputs()
    cut = lambda a: [a[lo:hi] for lo, hi in zip(offsets[:-1], offsets[1:])]  # noqa: E731

    def fresh():
        return MCSamples(samples=[c.copy() for c in cut(samples)], weights=[c.copy() for c in cut(weights)],
                         loglikes=[c.copy() for c in cut(loglikes)], names=names, ranges={"m3": (-3.0, None)})

    def state(mc, tag, out):
        mc.updateBaseStatistics()
        out[tag + "/numrows"] = np.int64(mc.numrows)
        out[tag + "/norm"] = np.float64(mc.norm)
        out[tag + "/means"] = mc.means.copy()
        out[tag + "/cov"] = mc.fullcov.copy()
        out[tag + "/max_mult"] = np.float64(mc.max_mult)
        if mc.loglikes is not None:
            out[tag + "/loglike_sum"] = np.float64(np.dot(mc.weights, mc.loglikes))
        if getattr(mc, "chain_offsets", None) is not None:
            out[tag + "/chain_offsets"] = np.asarray(mc.chain_offsets, dtype=np.int64)

    out = {}
    mc = fresh()
    state(mc, "base", out)
    ls = mc.getLikeStats()
    out["likestats/scalars"] = np.array([ls.logLike_sample, ls.logMeanInvLike if ls.logMeanInvLike is not None else np.nan,
                                         ls.meanLogLike, ls.logMeanLike, ls.complexity, ls.varLogLike])
    out["likestats/ND_bot"] = np.array([p.ND_limit_bot for p in mc.paramNames.names])
    out["likestats/ND_top"] = np.array([p.ND_limit_top for p in mc.paramNames.names])
    out["likestats/bestfit"] = np.array([p.bestfit_sample for p in mc.paramNames.names])
    # autocorrelation: many lags (the FFT route), both units, a vector argument, the length with and without corr=
    for j in (0, 3):
        for wu in (True, False):
            out["autocorr/%d/%d" % (j, wu)] = mc.getAutocorrelation(j, maxOff=700, weight_units=wu)
        out["corrlen/%d" % j] = np.float64(mc.getCorrelationLength(j))
    vec = mc.samples[:, 1] * mc.samples[:, 2]
    out["autocorr/vec"] = mc.getAutocorrelation(vec, maxOff=40, normalized=False)
    out["corrlen/given"] = np.float64(mc.getCorrelationLength(0, corr=out["autocorr/3/1"]))
    # a slowly mixing single chain: the 5 % crossing lies beyond a thousand lags
    rng = np.random.default_rng(5)
    n_ar = 60000
    e = rng.standard_normal(n_ar)
    ar = np.empty(n_ar)
    ar[0] = e[0]
    for t in range(1, n_ar):
        ar[t] = 0.9985 * ar[t - 1] + e[t]
    slow = MCSamples(samples=ar.reshape(-1, 1), names=["s"])
    out["slow/corrlen"] = np.float64(slow.getCorrelationLength(0))
    out["slow/corrlen_rows"] = np.float64(slow.getCorrelationLength(0, weight_units=False))
    mc = fresh(); mc.thin(3); state(mc, "thin3", out)
    mc = fresh(); mc.weighted_thin(2); state(mc, "wthin2", out)
    mc = fresh(); mc.filter(mc.samples[:, 0] > -0.4); state(mc, "filter", out)
    mc = fresh(); mc.reweightAddingLogLikes(extra.copy()); state(mc, "reweight", out)
    mc = fresh(); mc.cool(1.7); state(mc, "cool", out)
    mc = fresh(); mc.removeBurn(0.2); state(mc, "burn", out)
    mc = fresh()
    mc.addDerived(mc.samples[:, 0] * mc.samples[:, 1] + 0.2 * mc.samples[:, 3], "d01", label="d_{01}", range=(None, 6.0))
    state(mc, "derived", out)
    out["derived/P1d"] = mc.get1DDensity("d01").P
    out["derived/P2d_sum"] = np.float64(np.sum(mc.get2DDensity("m0", "d01").P))
    out["derived/isDerived"] = np.bool_(mc.paramNames.parWithName("d01").isDerived)
    fx = np.column_stack([samples[:, 0], np.full(len(samples), 0.25), samples[:, 1]])
    mc = MCSamples(samples=fx, weights=weights.copy(), names=["a", "fixed", "b"])
    out["fixed/names_after"] = np.array([p.name for p in mc.paramNames.names])
    out["fixed/value"] = np.float64(mc.ranges.getLower("fixed"))
    # ---- convolve.py on seeded inputs
    rng = np.random.default_rng(123)
    x1, y1, ys, x2, y2 = rng.random(1024), rng.random(141), rng.random(1203), rng.random((128, 128)), rng.random((31, 31))
    xl = rng.random(1500)
    for mode in ("same", "valid", "full"):
        out["conv1d/direct/" + mode] = rconv.convolve1D(x1, y1, mode)
        out["conv1d/fft/" + mode] = rconv.convolve1D(xl, ys, mode, largest_size=3000)
        out["conv2d/" + mode] = rconv.convolve2D(x2, y2, mode, largest_size=128 + 2 * 15 + 31)
    out["conv1d/periodic"] = rconv.convolve1D(x1, y1, "periodic")
    for mode in ("periodic", "periodic_x", "periodic_y"):
        out["conv2d/" + mode] = rconv.convolve2D(x2[:96, :80], y2[:21, :17], mode)
    z = rng.standard_normal(20000)
    out["autoconv/norm"] = rconv.autoConvolve(z, 300)
    out["autoconv/raw"] = rconv.autoConvolve(z, 300, normalize=False)
    out["autocorrfn"] = rconv.autoCorrelation(z, 200)
    return out


def main():
    if "--mask-corners" in sys.argv:
        out = golden_mask_corners({fx["name"]: fx for fx in fixture_zoo()})
        path = os.path.join(HERE, "mask_function_corners.npz")
        np.savez_compressed(path, **out)
        print("mask corners:", len(out), "arrays", os.path.getsize(path) // 1024, "KiB")
        return
    if "--mutators" in sys.argv:
        out = golden_mutators()
        path = os.path.join(HERE, "mutators.npz")
        np.savez_compressed(path, **out)
        print("mutators:", len(out), "arrays", os.path.getsize(path) // 1024, "KiB")
        return
    if "--reference-unit-tests" in sys.argv:
        out = golden_reference_unit_tests()
        path = os.path.join(HERE, "reference_unit_tests.npz")
        np.savez_compressed(path, **out)
        print("reference unit tests:", {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()},
              os.path.getsize(path) // 1024, "KiB")
        return
    if "--fixtures" in sys.argv:  # only the named fixture files (e.g. when the zoo grows)
        wanted = sys.argv[sys.argv.index("--fixtures") + 1].split(",")
        zoo = {fx["name"]: fx for fx in fixture_zoo()}
        for nm in wanted:
            out = golden_for_fixture(**zoo[nm])
            path = os.path.join(HERE, "fixture_%s.npz" % nm)
            np.savez_compressed(path, **out)
            print(nm, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")
            if "--margestats" in sys.argv:
                np.savez_compressed(os.path.join(HERE, "margestats_%s.npz" % nm), **golden_margestats(zoo[nm]))
        return
    xs = np.unique(np.concatenate([np.arange(1, 3000), np.geomspace(3000, 2.0e9, 1500).astype(np.int64)]))
    np.savez_compressed(os.path.join(HERE, "fftnumbers.npz"), x=xs, y=nearestFFTnumber(xs))
    np.savez_compressed(os.path.join(HERE, "convergence.npz"), **golden_convergence())
    zoo = {fx["name"]: fx for fx in fixture_zoo()}
    for nm in ("shapes", "c1_bounded", "block10_weighted"):
        np.savez_compressed(os.path.join(HERE, "margestats_%s.npz" % nm), **golden_margestats(zoo[nm]))
    np.savez_compressed(os.path.join(HERE, "splittests.npz"),
                        **{nm: golden_split_tests(zoo[nm]) for nm in ("shapes_intweights", "c1_bounded", "block10_weighted")})
    np.savez_compressed(os.path.join(HERE, "mask_function.npz"), **golden_mask_function(zoo))
    if "--only-mask" in sys.argv:
        return
    np.savez_compressed(os.path.join(HERE, "raftery_lewis.npz"), **golden_raftery_lewis())
    if "--only-rl" in sys.argv:
        return
    np.savez_compressed(os.path.join(HERE, "nd_ranges.npz"), **golden_nd_ranges(zoo))
    if "--only-nd" in sys.argv:
        return
    np.savez_compressed(os.path.join(HERE, "meanlikes.npz"), **golden_meanlikes(zoo))
    if "--only-new" in sys.argv:
        return
    for fx in zoo.values():
        out = golden_for_fixture(**fx)
        path = os.path.join(HERE, "fixture_%s.npz" % fx["name"])
        np.savez_compressed(path, **out)
        print(fx["name"], len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
