"""Seeded inputs shared by make_golden.py (which runs the reference on them) and the tests (which run the build)."""
import numpy as np


def mutator_inputs():
    from oracle.fixtures import mcmc_chains_fixture

    samples, weights, loglikes, names, offsets = mcmc_chains_fixture(nchains=3, N=2500, n=4)
    rng = np.random.default_rng(77)
    extra = 0.3 * rng.standard_normal(len(weights)) ** 2  # the log-likelihoods added by reweightAddingLogLikes
    return samples, weights, loglikes, names, offsets, extra
