"""
Large-sample pinning of the device solvers' source (getdist_amd/csrc/solvers.hpp, compiled for the host by
tests/native/build.py) against scipy -- the long version of tests/test_native_solvers.py:

  * TNC: evaluation-point sequences, results, return codes of scipy.optimize.minimize(method="TNC") vs the port on
    N random AMISE problems (2 and 3 unknowns), normal and wide parameter ranges;
  * get_h: the whole bandwidth recipe in C++ vs the oracle (scipy), bit for bit;
  * hybrd1 / brentq: the 1D ISJ recipe vs scipy fsolve / brentq on the histogram shape zoo.

    python scripts/validate_native_solvers.py [n_tuples]      (default 15000; prints one summary line per block)
"""
import ctypes
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "native"))

import test_native_solvers as T  # noqa: E402
from oracle import kde_oracle as ko  # noqa: E402
from oracle.fixtures import histogram_shape_zoo, random_psi_tuples  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
    lib = T.lib.__wrapped__() if hasattr(T.lib, "__wrapped__") else None
    if lib is None:
        import build

        lib = build.load()
        pd, pi = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        lib.gdt_hybrd1.argtypes = [T.FCN, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, pd, pi]
        lib.gdt_brentq.argtypes = [T.FCN, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, pd, pi]
        lib.gdt_tnc.argtypes = [T.FCN_ND, ctypes.c_int, pd, pd, pd, pd, pi, pi, pi]
        lib.gdt_get_h.argtypes = [pd, ctypes.c_double, ctypes.c_double, ctypes.c_int, pd, pi]
    t0 = time.time()
    runs = bad = aborted = 0
    codes = {}
    for wide in (False, True):
        for psi, N, corr in random_psi_tuples(n // 2, seed=101 + wide, wide=wide):
            p = np.zeros((5, 5))
            p[0, 4], p[4, 0], p[2, 2], p[0, 0], p[1, 3], p[3, 1] = psi
            h_x, h_y, _ = ko.get_h_from_psi(psi, N, 0.0, False)
            todo = [([h_x, h_y, corr], [(0.001, 0.3), (0.001, 0.3), (-0.99, 0.99)], None)]
            if corr:
                todo.append((list(np.array([h_x, h_y]) / np.sqrt(1 - abs(corr))), [(0.001, 0.3), (0.001, 0.3)], corr))
            for x0, bounds, c in todo:
                ref, seq_ref, (x, rc, success, nit), seq = T._tnc_both(lib, p, N, x0, bounds, c)
                runs += 1
                if ref is None:
                    aborted += 1
                    ok = rc == 7 and seq == seq_ref
                else:
                    ok = (seq_ref[:len(seq)] == seq and len(seq_ref) - len(seq) in (0, 1 + len(x0)) and np.array_equal(ref.x, x)
                          and ref.status == rc and bool(ref.success) == success and ref.nit == nit)
                    codes[rc] = codes.get(rc, 0) + 1
                bad += not ok
    print("TNC: %d runs, %d mismatches (evaluation sequence / result / code), %d aborted runs matched, return codes %s, %.0f s"
          % (runs, bad, aborted, dict(sorted(codes.items())), time.time() - t0))
    t0 = time.time()
    tot = diff = 0
    for psi, N, corr in random_psi_tuples(n, seed=202):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = tuple(float(v) for v in ko.get_h_from_psi(tuple(np.float64(v) for v in psi), N, corr, True))
        out, nf = (ctypes.c_double * 3)(), ctypes.c_int()
        st = lib.gdt_get_h((ctypes.c_double * 6)(*psi), N, corr, 1, out, ctypes.byref(nf))
        tot += 1
        diff += not (st == 0 and tuple(out) == want)
    print("get_h: %d psi tuples, %d not bit-identical to the scipy path, %.0f s" % (tot, diff, time.time() - t0))
    t0 = time.time()
    tot = diff = 0
    for kind, hist, neff in histogram_shape_zoo(800, seed=77):
        fp = T.functional_of(hist, neff)
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
        info = lib.gdt_hybrd1(T.FCN(cb), h0, h0 / 20, 400, 1.0, ctypes.byref(x_out), ctypes.byref(nfev))
        got = None if info < 0 else x_out.value
        if got is not None and got < 0.019 * n_scaling and got / 20 > 0:
            st = lib.gdt_brentq(T.FCN(cb), 0.019 * n_scaling, 0.5, got / 20, 4 * np.finfo(float).eps, 100,
                                ctypes.byref(x_out), ctypes.byref(nfev))
            if st == 0:
                got = x_out.value
        tot += 1
        diff += got != want
    print("1D ISJ (hybrd1 + brentq): %d histogram shapes, %d not bit-identical to scipy fsolve/brentq, %.0f s"
          % (tot, diff, time.time() - t0))
    return 1 if (bad or diff) else 0


if __name__ == "__main__":
    sys.exit(main())
