"""
CPU tests of the scalar solvers that run inside the HIP kernels (getdist_amd/csrc/solvers.hpp), compiled for the host
by tests/native/build.py and driven through ctypes with a PYTHON callback as the function.  scipy's own solver and the
port then see bit-identical function values, so the sequences of evaluation points and the results must be
bit-identical too: this pins the iteration path, which defines the reference's answer.
"""

import ctypes
import os
import sys
import warnings

import numpy as np
import pytest
from scipy import fftpack
from scipy.optimize import brentq, fsolve, minimize

from oracle import kde_oracle as ko
from oracle.fixtures import histogram_shape_zoo as shape_zoo

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "native"))

FCN = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_int))
FCN_ND = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_int))


@pytest.fixture(scope="module")
def lib():
    import build

    lib = build.load()
    pd, pi = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lib.gdt_hybrd1.argtypes = [FCN, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, pd, pi]
    lib.gdt_brentq.argtypes = [FCN, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, pd, pi]
    lib.gdt_tnc.argtypes = [FCN_ND, ctypes.c_int, pd, pd, pd, pd, pi, pi, pi]
    lib.gdt_get_h.argtypes = [pd, ctypes.c_double, ctypes.c_double, ctypes.c_int, pd, pi]
    return lib


def functional_of(hist, neff):
    I = np.arange(1, hist.size) ** 2
    logI = np.log(I)
    a2 = (fftpack.dct(hist / np.sum(hist))[1:] / 2) ** 2
    return lambda h: ko.isj_fixed_point(h, neff, I, logI, a2)


def test_hybrd1_follows_scipy_fsolve_evaluation_by_evaluation(lib):
    infos = set()
    for kind, hist, neff in shape_zoo(96):
        fp = functional_of(hist, neff)
        ref_seq, my_seq = [], []

        def logged(h):
            ref_seq.append(float(np.atleast_1d(h)[0]))
            return fp(h)

        h0 = 0.53 * neff ** (-1.0 / 5)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                ref, _, ier, _ = fsolve(logged, h0, xtol=h0 / 20, factor=1, full_output=True)
            except Exception:  # "zero f": the functional underflowed inside the solver
                ref = None

            def cb(x, fail):
                my_seq.append(x)
                try:
                    return float(np.atleast_1d(fp(np.array([x])))[0])
                except Exception:
                    fail[0] = 1
                    return 0.0

            x_out, nfev = ctypes.c_double(), ctypes.c_int()
            info = lib.gdt_hybrd1(FCN(cb), h0, h0 / 20, 400, 1.0, ctypes.byref(x_out), ctypes.byref(nfev))
        # scipy evaluates x0 twice more than MINPACK itself (shape checks of the wrapper)
        if ref is None:  # raised in the wrapper's first shape-check call already
            assert info == -1 and ref_seq[-1:] == my_seq[-1:]
        else:
            assert ref_seq[2:] == my_seq, (kind, ref_seq, my_seq)
            assert x_out.value == ref[0] and info == ier and nfev.value == len(my_seq)
        infos.add(info)
    assert {1, 5} <= infos  # converged runs and slow-progress exits (flat shapes) were both exercised


def test_brentq_and_the_whole_1d_bandwidth_recipe(lib):
    """kde_bandwidth.py:113-135 with the ported solvers equals the oracle (scipy solvers) bit for bit, including the
    cases that enter the brentq re-check and those where brentq refuses the bracket."""
    rechecks = refused = 0
    for kind, hist, neff in shape_zoo(64, seed=11):
        fp = functional_of(hist, neff)
        want = ko.isj_bandwidth_binned(hist, neff)

        def cb(x, fail):
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    return float(np.atleast_1d(fp(np.array([x])))[0])
            except Exception:
                fail[0] = 1
                return 0.0

        n_scaling = neff ** (-1.0 / 5)
        h0 = 0.53 * n_scaling
        x_out, nfev = ctypes.c_double(), ctypes.c_int()
        info = lib.gdt_hybrd1(FCN(cb), h0, h0 / 20, 400, 1.0, ctypes.byref(x_out), ctypes.byref(nfev))
        got = None if info < 0 else x_out.value
        if got is not None and got < 0.019 * n_scaling and got / 20 > 0:
            rechecks += 1
            st = lib.gdt_brentq(FCN(cb), 0.019 * n_scaling, 0.5, got / 20, 4 * np.finfo(float).eps, 100,
                                ctypes.byref(x_out), ctypes.byref(nfev))
            if st == 0:
                got = x_out.value
            else:
                refused += 1
        assert got == want, (kind, got, want)
    assert rechecks > 0


def test_brentq_port_matches_scipy_on_smooth_functions(lib):
    rng = np.random.default_rng(3)
    for _ in range(200):
        c = rng.uniform(0.05, 0.95)
        p = rng.integers(1, 6)
        f = lambda x: (x - c) ** p * np.sign(x - c) ** (p + 1) + 0.3 * np.sin(5 * (x - c))  # noqa: E731
        seq_ref, seq = [], []

        def logged(x):
            seq_ref.append(x)
            return f(x)

        def cb(x, fail):
            seq.append(x)
            return float(f(x))

        xtol = 10 ** rng.uniform(-12, -3)
        try:
            want = brentq(logged, 0.0, 1.0, xtol=xtol)
        except ValueError:
            want = None
        x_out, nfev = ctypes.c_double(), ctypes.c_int()
        st = lib.gdt_brentq(FCN(cb), 0.0, 1.0, xtol, 4 * np.finfo(float).eps, 100, ctypes.byref(x_out), ctypes.byref(nfev))
        if want is None:
            assert st == -1
        else:
            assert st == 0 and x_out.value == want and seq == seq_ref


def _tnc_both(lib, p, N, x0, bounds, corr):
    """Run scipy's TNC and the port on the same Python AMISE; returns (reference result or None, its evaluation points,
    port's (x, rc, success, nit), its evaluation points)."""
    seq_ref, seq = [], []

    def f_ref(x, *a):
        seq_ref.append(tuple(float(v) for v in x))
        return ko.amise_from_psi(x, p, N, corr)

    ref = None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = minimize(f_ref, np.array(x0), method="TNC", bounds=bounds)
    except Exception:
        pass

    def cb(xp, n, fail):
        x = np.array([xp[i] for i in range(n)])
        seq.append(tuple(float(v) for v in x))
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return float(ko.amise_from_psi(x, p, N, corr))
        except Exception:
            fail[0] = 1
            return 0.0

    n = len(x0)
    arr = ctypes.c_double * n
    x_out, suc, nf, nit = arr(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib.gdt_tnc(FCN_ND(cb), n, arr(*x0), arr(*[b[0] for b in bounds]), arr(*[b[1] for b in bounds]), x_out,
                     ctypes.byref(suc), ctypes.byref(nf), ctypes.byref(nit))
    return ref, seq_ref, (np.array(list(x_out)), rc, bool(suc.value), nit.value), seq


def test_tnc_port_follows_scipy_evaluation_by_evaluation(lib):
    """scipy.optimize.minimize(method="TNC") with finite-difference gradients and bounds, as get_h calls it
    (kde_bandwidth.py:276-299), against the port in csrc/solvers.hpp: identical sequences of evaluation points, final
    iterate, return code, iteration count -- also where the callback raises ("bias not positive definite"), where the
    start point violates the bounds, where a bound becomes active and is released, and where the line search fails.
    (scripts/validate_native_solvers.py repeats this on 3 x 10^4 runs over wider parameter ranges.)"""
    from oracle.fixtures import random_psi_tuples

    n_runs = n_abort = n_bound = 0
    codes = set()
    for wide in (False, True):
        for psi, N, corr in random_psi_tuples(220, seed=7 + wide, wide=wide):
            p = np.zeros((5, 5))
            p[0, 4], p[4, 0], p[2, 2], p[0, 0], p[1, 3], p[3, 1] = psi
            h_x, h_y, _ = ko.get_h_from_psi(psi, N, 0.0, False)
            runs = [([h_x, h_y, corr], [(0.001, 0.3), (0.001, 0.3), (-0.99, 0.99)], None)]
            if corr:
                runs.append((list(np.array([h_x, h_y]) / np.sqrt(1 - abs(corr))), [(0.001, 0.3), (0.001, 0.3)], corr))
            for x0, bounds, c in runs:
                ref, seq_ref, (x, rc, success, nit), seq = _tnc_both(lib, p, N, x0, bounds, c)
                n_runs += 1
                if ref is None:
                    n_abort += 1
                    assert rc == 7 and seq == seq_ref, (psi, N, corr)
                    continue
                # scipy re-evaluates f and the gradient at the returned x (memoised if it was the last point)
                assert seq_ref[:len(seq)] == seq and len(seq_ref) - len(seq) in (0, 1 + len(x0)), (psi, N, corr)
                assert np.array_equal(ref.x, x) and ref.status == rc and bool(ref.success) == success and ref.nit == nit
                codes.add(rc)
                lo, hi = np.array(bounds).T
                n_bound += bool(np.any(np.minimum(np.abs(x - lo), np.abs(x - hi)) < 1e-12))
    assert n_runs > 700 and n_abort > 0 and n_bound > 0 and {1, 4} <= codes, (n_runs, n_abort, n_bound, codes)


def test_get_h_port_equals_the_oracle_bit_for_bit(lib):
    """KernelOptimizer2D.get_h in plain C++ (the code the kernel runs: closed forms, AMISE, two TNC runs, acceptance
    rules) against the oracle's scipy version on random psi tuples: identical doubles."""
    from oracle.fixtures import random_psi_tuples

    kinds = [0, 0, 0]
    for psi, N, corr in random_psi_tuples(1200, seed=3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = tuple(float(v) for v in ko.get_h_from_psi(tuple(np.float64(v) for v in psi), N, corr, True))
        out, nf = (ctypes.c_double * 3)(), ctypes.c_int()
        st = lib.gdt_get_h((ctypes.c_double * 6)(*psi), N, corr, 1, out, ctypes.byref(nf))
        assert st == 0 and tuple(out) == want, (psi, N, corr, tuple(out), want)
        kinds[0 if want[2] == 0 else (1 if want[2] == corr else 2)] += 1
        want0 = tuple(float(v) for v in ko.get_h_from_psi(psi, N, corr, False))
        lib.gdt_get_h((ctypes.c_double * 6)(*psi), N, corr, 0, out, ctypes.byref(nf))
        assert tuple(out) == want0
    assert kinds[0] > 0 and kinds[2] > 0  # both "closed form kept" and "TNC result accepted" occur
