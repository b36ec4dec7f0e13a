"""
Seeded synthetic sample sets for the configs in BASELINE.json.

Pure numpy (+ scipy.special.ndtri for Gaussian quantiles); identical output on any box.  Rows are
generated in 1M-row chunks, each from its own child of ``SeedSequence(20260926)``, so the data do
not depend on how many worker threads generate them.

Columns are produced *columns-first*: the returned ``samples`` is an (N, n) Fortran-ordered view,
so ``samples[:, j]`` is contiguous and uploads to the device SoA layout without a transpose.
"""

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from scipy.special import ndtri

BASE_SEED = 20260926
CHUNK = 1_000_000
RHO_CYCLE = (0.0, 0.3, 0.6, 0.9, 0.97, 0.995, -0.3, -0.6, -0.9, 0.0)
# parameter index -> (lower quantile or None, upper quantile or None); SURVEY.md section 8d
BOUND_QUANTILES = {4: (0.2, None), 9: (0.2, None), 19: (0.2, None), 18: (0.2, None), 14: (0.3, None),
                   24: (None, 0.8), 29: (None, 0.8), 34: (None, 0.7), 39: (0.1, 0.9), 49: (0.05, 0.95)}


def _chunks(N):
    return [(a, min(a + CHUNK, N)) for a in range(0, N, CHUNK)]


def _run_chunks(fn, N, stream):
    seeds = np.random.SeedSequence([BASE_SEED, stream]).spawn(len(_chunks(N)))
    jobs = list(zip(_chunks(N), seeds))
    workers = max(1, min(16, os.cpu_count() or 1))
    if workers == 1 or len(jobs) == 1:
        for (a, b), s in jobs:
            fn(a, b, np.random.default_rng(s))
    else:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(lambda job: fn(job[0][0], job[0][1], np.random.default_rng(job[1])), jobs))


def block_recipe(n, N, weighted=False, stream=1):
    """
    The block recipe: parameters in blocks of 5; block b is a zero-mean Gaussian with sigma_i ~ U(0.5, 2)
    and uniform in-block correlation RHO_CYCLE[b % 10] (negative: alternating member signs); blocks are
    independent; block 0 is bimodal (35% of rows shifted by +3 sigma in all its coordinates); hard bounds
    by reflection at the Gaussian quantiles in BOUND_QUANTILES, also returned as ``ranges``.

    :return: samples (N, n) F-ordered float64, weights (N,) or None, names, ranges {name: (lo, hi)}
    """
    nblocks = (n + 4) // 5
    sig = np.random.default_rng(np.random.SeedSequence([BASE_SEED, 0])).uniform(0.5, 2.0, size=nblocks * 5)
    cols = np.empty((n, N), dtype=np.float64)
    weights = np.empty(N, dtype=np.float64) if weighted else None
    bounds = {}
    for p, (ql, qu) in BOUND_QUANTILES.items():
        if p < n and p >= 5:  # block 0 is the mixture: leave it unbounded
            bounds[p] = (None if ql is None else sig[p] * ndtri(ql), None if qu is None else sig[p] * ndtri(qu))
    if 4 in BOUND_QUANTILES and n > 4:
        # p4 sits in the bimodal block: bound it at the 20% quantile of the un-shifted component
        bounds[4] = (sig[4] * ndtri(0.2), None)

    def fill(a, b, rng):
        m = b - a
        for blk in range(nblocks):
            rho = RHO_CYCLE[blk % len(RHO_CYCLE)]
            common = rng.standard_normal(m)
            shift = (rng.random(m) < 0.35) * 3.0 if blk == 0 else None
            for k in range(5):
                p = blk * 5 + k
                z = rng.standard_normal(m)
                if p >= n:
                    continue
                v = np.sqrt(abs(rho)) * common + np.sqrt(1 - abs(rho)) * z
                if rho < 0 and k % 2:
                    v = -v
                if shift is not None:
                    v += shift
                v *= sig[p]
                if p in bounds:
                    lo, hi = bounds[p]
                    if lo is not None:
                        v = np.where(v < lo, 2 * lo - v, v)
                    if hi is not None:
                        v = np.where(v > hi, 2 * hi - v, v)
                    if lo is not None and hi is not None:  # second reflection for far tails
                        v = np.where(v < lo, 2 * lo - v, v)
                        v = np.clip(v, lo, hi)
                cols[p, a:b] = v
        if weighted:
            weights[a:b] = rng.exponential(1.0, m)

    _run_chunks(fill, N, stream)
    names = ["p%d" % i for i in range(n)]
    ranges = {names[p]: bounds[p] for p in sorted(bounds)}
    return cols.T, weights, names, ranges


def config_c1(N=100_000, bounded=False):
    """C1: one 4-parameter correlated Gaussian, unit weights (variant: last parameter reflected at 0)."""
    sig = np.array([1.0, 2.0, 0.5, 3.0])
    corr = np.array([[1, 0.9, 0.3, 0], [0.9, 1, 0.2, 0], [0.3, 0.2, 1, -0.6], [0, 0, -0.6, 1]], dtype=float)
    L = np.linalg.cholesky(corr * np.outer(sig, sig))
    cols = np.empty((4, N))

    def fill(a, b, rng):
        z = rng.standard_normal((4, b - a))
        cols[:, a:b] = L @ z

    _run_chunks(fill, N, stream=2)
    names = ["a", "b", "c", "d"]
    ranges = {}
    if bounded:
        cols[3] = np.abs(cols[3])
        ranges = {"d": (0.0, None)}
    return cols.T, None, names, ranges


def config_c2(N=10_000_000):
    """C2: first 30 parameters of the block recipe, w ~ Exp(1)."""
    return block_recipe(30, N, weighted=True, stream=3)


def config_c3(N=10_000_000, n=50):
    """C3 (headline): 50 parameters, unit weights, 1225 pairs."""
    return block_recipe(n, N, weighted=False, stream=4)


def config_c4(nchains=8, N=5_000_000, n=100):
    """C4: nchains chains of one n-dim Gaussian with random SPD covariance, w ~ Exp(1)."""
    rng0 = np.random.default_rng(np.random.SeedSequence([BASE_SEED, 5]))
    A = rng0.standard_normal((n, n))
    L = np.linalg.cholesky(A @ A.T / n + 0.1 * np.eye(n))
    total = nchains * N
    cols = np.empty((n, total))
    weights = np.empty(total)

    def fill(a, b, rng):
        cols[:, a:b] = L @ rng.standard_normal((n, b - a))
        weights[a:b] = rng.exponential(1.0, b - a)

    _run_chunks(fill, total, stream=6)
    offsets = np.arange(nchains + 1) * N
    return cols.T, weights, ["p%d" % i for i in range(n)], offsets


def config_c4_chain(chain, N=5_000_000, n=100):
    """One chain of the C4 recipe, generated independently of the others (chain-per-GPU runs: each rank makes its own)."""
    rng0 = np.random.default_rng(np.random.SeedSequence([BASE_SEED, 5]))
    A = rng0.standard_normal((n, n))
    L = np.linalg.cholesky(A @ A.T / n + 0.1 * np.eye(n))
    cols = np.empty((n, N))
    weights = np.empty(N)

    def fill(a, b, rng):
        cols[:, a:b] = L @ rng.standard_normal((n, b - a))
        weights[a:b] = rng.exponential(1.0, b - a)

    _run_chunks(fill, N, stream=600 + chain)
    return cols.T, weights, ["p%d" % i for i in range(n)]


def triangle_pairs(n):
    """Lower-triangle (x, y) index pairs in the order the reference's triangle plot visits them
    (plots.py:2845-2878: for each row i2>i, x=param i, y=param i2)."""
    return [(i, i2) for i in range(n) for i2 in range(i + 1, n)]
