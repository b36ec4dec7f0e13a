"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the Raftery-Lewis and CorrSteps blocks of
MCSamples.getConvergeTests (getdist/mcsamples.py:1039-1221) and of the weight-one thinning they rest on
(getdist/chains.py:878-916), over plain arrays.  Parity PINNED: oracle/validate_against_reference.py compares it with
the imported reference (thin indices bit-equal on random weights; the Raftery-Lewis table and the CorrSteps numbers
against the reference's own text output and primitives); tests/golden/raftery_lewis.npz holds the reference outputs.
"""

import math

import numpy as np


def thin_indices(factor, weights):
    """
    chains.py:878-916 thin_indices_single_samples.  With C_i the inclusive cumulative (integer) weight:
      factor >= max(weights): np.unique(C // factor, return_index=True)[1]  (:889-892)
      otherwise: row i is emitted once per multiple of `factor` in (C_{i-1}, C_i]  (the while loop :894-914),
    which np.repeat states without the Python loop.
    """
    weights = np.asarray(weights)
    norm1 = np.sum(weights)
    w = weights.astype(int)
    if abs(np.sum(w) - norm1) > 1e-4:
        raise ValueError("Can only thin with integer weights")
    if factor != int(factor):
        raise ValueError("Thin factor must be integer")
    factor = int(factor)
    v = np.cumsum(w) // factor
    if factor >= np.max(w):
        return np.unique(v, return_index=True)[1]
    return np.repeat(np.arange(len(w)), np.diff(v, prepend=0))


def _confidence(x, w, fracs):
    """chains.py:793-838 on one chain: x[idx[min(searchsorted(cumsum(w[idx]), norm*f), N-1)]]"""
    idx = x.argsort()
    cum = np.cumsum(w[idx])
    ix = np.searchsorted(cum, np.sum(w) * np.atleast_1d(fracs))
    return x[idx[np.minimum(ix, len(idx) - 1)]]


def raftery_lewis(samples, weights, chain_offsets, nparamMC, test_confidence=0.95):
    """
    mcsamples.py:1039-1165.  Returns dict(markov_thin, thin_fac (= indep thin), nburn) per chain, or None where the
    reference returns early ("Raftery and Lewis estimator had problems", :1134-1136).
    """
    samples = np.asarray(samples)
    weights = np.asarray(weights, dtype=np.float64)
    chains = [(samples[a:b], weights[a:b]) for a, b in zip(chain_offsets[:-1], chain_offsets[1:])]
    nc = len(chains)
    limits = np.array([1 - (1 - test_confidence) / 2, (1 - test_confidence) / 2])
    thin_fac = np.empty(nc, dtype=int)
    epsilon = 0.001
    nburn = np.zeros(nc, dtype=int)
    markov_thin = np.zeros(nc, dtype=int)
    hardest, hardestend = -1, 0  # deliberately NOT reset per chain (as in the reference)

    class LoopException(Exception):
        pass

    for ix, (cs, cw) in enumerate(chains):
        thin_fac[ix] = int(round(np.max(cw)))
        thin_rows = None
        try:
            for j in range(nparamMC):
                confids = _confidence(cs[:, j], cw, limits)
                for endb in (0, 1):
                    u = confids[endb]
                    while True:
                        thin_ix = thin_indices(thin_fac[ix], cw)
                        thin_rows = len(thin_ix)
                        if thin_rows < 2:
                            break
                        binchain = np.ones(thin_rows, dtype=int)
                        binchain[cs[thin_ix, j] >= u] = 0
                        tran = np.bincount(binchain[:-2] * 4 + binchain[1:-1] * 2 + binchain[2:], minlength=8).reshape((2, 2, 2))
                        g2 = 0
                        for i1 in (0, 1):
                            for i2 in (0, 1):
                                for i3 in (0, 1):
                                    if tran[i1][i2][i3] != 0:
                                        fitted = float((tran[i1][i2][0] + tran[i1][i2][1]) * (tran[0][i2][i3] + tran[1][i2][i3])) \
                                            / float(tran[0][i2][0] + tran[0][i2][1] + tran[1][i2][0] + tran[1][i2][1])
                                        focus = float(tran[i1][i2][i3])
                                        g2 += math.log(focus / fitted) * focus
                        g2 *= 2
                        if g2 - math.log(float(thin_rows - 2)) * 2 < 0:
                            break
                        thin_fac[ix] += 1
                    if np.sum(tran[:, 0, 1]) == 0 or np.sum(tran[:, 1, 0]) == 0:
                        thin_fac[ix] = 0
                        raise LoopException()
                    alpha = np.sum(tran[:, 0, 1]) / float(np.sum(tran[:, 0, 0]) + np.sum(tran[:, 0, 1]))
                    beta = np.sum(tran[:, 1, 0]) / float(np.sum(tran[:, 1, 0]) + np.sum(tran[:, 1, 1]))
                    probsum = alpha + beta
                    tmp1 = math.log(probsum * epsilon / max(alpha, beta)) / math.log(abs(1.0 - probsum))
                    if int(tmp1 + 1) * thin_fac[ix] > nburn[ix]:
                        nburn[ix] = int(tmp1 + 1) * thin_fac[ix]
                        hardest = j
                        hardestend = endb
            markov_thin[ix] = thin_fac[ix]
            hardest = max(hardest, 0)
            u = _confidence(samples[:, hardest], weights, (1 - test_confidence) / 2 if hardestend != 0
                            else 1 - (1 - test_confidence) / 2)[0]
            while True:
                thin_ix = thin_indices(thin_fac[ix], cw)
                thin_rows = len(thin_ix)
                if thin_rows < 2:
                    break
                binchain = np.ones(thin_rows, dtype=int)
                binchain[cs[thin_ix, hardest] >= u] = 0
                tran2 = np.bincount(binchain[:-1] * 2 + binchain[1:], minlength=4).reshape(2, 2)
                g2 = 0
                for i1 in (0, 1):
                    for i2 in (0, 1):
                        if tran2[i1][i2] != 0:
                            fitted = float((tran2[i1][0] + tran2[i1][1]) * (tran2[0][i2] + tran2[1][i2])) / float(thin_rows - 1)
                            focus = float(tran2[i1][i2])
                            if fitted <= 0 or focus <= 0:
                                return None
                            g2 += np.log(focus / fitted) * focus
                g2 *= 2
                if g2 - np.log(float(thin_rows - 1)) < 0:
                    break
                thin_fac[ix] += 1
        except LoopException:
            pass
        except Exception:  # the reference's bare `except:` (:1146-1147)
            thin_fac[ix] = 0
        if thin_fac[ix] and thin_rows < 2:
            thin_fac[ix] = 0
    return dict(markov_thin=markov_thin, thin_fac=thin_fac, nburn=nburn)


def corr_steps(samples, weights, chain_offsets, vars_, autocorr_thin, corr_length_steps=15):
    """mcsamples.py:1183-1210: parameter auto-correlations of the thinned chains as a function of step separation."""
    samples = np.asarray(samples)
    weights = np.asarray(weights, dtype=np.float64)
    chains = [(samples[a:b], weights[a:b]) for a, b in zip(chain_offsets[:-1], chain_offsets[1:])]
    nparam = samples.shape[1]
    thin_rows = len(thin_indices(autocorr_thin, weights))
    maxoff = int(min(corr_length_steps, thin_rows // (2 * len(chains))))
    if maxoff <= 0:
        return None
    corrs = np.zeros([maxoff, nparam])
    for cs, cw in chains:
        thin_ix = thin_indices(autocorr_thin, cw)
        thin_rows = len(thin_ix)
        maxoff = min(maxoff, thin_rows // autocorr_thin)
        means = cw.dot(cs) / np.sum(cw)
        for j in range(nparam):
            diff = (cs[:, j] - means[j])[thin_ix]
            for off in range(1, maxoff + 1):
                corrs[off - 1][j] += np.dot(diff[off:], diff[:-off]) / (thin_rows - off) / vars_[j]
    corrs /= len(chains)
    return corrs[:maxoff]
