"""
CPU tests of the host-side scalar solvers kept in the product (getdist_amd/mcsamples.py): given the SAME inputs
as the oracle they must return the oracle's result bit-for-bit -- this is the deterministic half of the
solver-path parity argument (the chaotic half is documented in DESIGN.md).
"""

import numpy as np

from getdist_amd import mcsamples as hm
from oracle import kde_oracle as ko


def test_get_h_matches_oracle(zoo):
    fx = zoo["block10_weighted"]
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    n_tnc = 0
    for (a, b) in fx["pairs"]:
        d = orc.density_2d(a, b)
        for corr, do_corr in ((0.0, True), (0.15, True), (-0.4, True), (0.3, False)):
            tr = {}
            opt = ko.Optimizer2D(d["histbins"], 4000.0, corr, do_correlation=do_corr, fallback_t=1e-4, trace=tr)
            want = opt.get_h()
            psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr.get("p_00", np.nan), tr.get("p_13", np.nan),
                   tr.get("p_31", np.nan))
            got = hm._get_h(psi, 4000.0, corr, do_corr)
            assert tuple(map(float, got)) == tuple(map(float, want)), (a, b, corr, got, want)
            n_tnc += do_corr
    assert n_tnc > 0


def test_cov_to_corr_and_bounds():
    c = np.array([[4.0, 1.0], [1.0, 9.0]])
    assert np.allclose(hm.covToCorr(c), ko.cov_to_corr(c))
    b = hm.ParamBounds()
    b.setRange("x", (0, None))
    b.setRange("phi", (0, 6.28, "periodic"))
    assert b.getLower("x") == 0.0 and b.getUpper("x") is None and "phi" in b.periodic


def test_tnc_pool_matches_serial_and_is_safe_without_main_guard(tmp_path):
    """The worker pool gives the serial results, and a driver script WITHOUT a __main__ guard is not re-executed."""
    import subprocess
    import sys
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "noguard.py"
    script.write_text(
        "import sys, os\n"
        "sys.path.insert(0, %r)\n"
        "open(os.path.join(%r, 'ran_%%d' %% os.getpid()), 'w').close()\n"
        "import numpy as np\n"
        "from getdist_amd import mcsamples as hm\n"
        "rng = np.random.default_rng(1)\n"
        "jobs = []\n"
        "for k in range(40):\n"
        "    p02, p20 = 50 + 10 * rng.random(), 60 + 10 * rng.random()\n"
        "    jobs.append(((np.float64(p02), np.float64(p20), np.float64(5 * rng.random()), np.float64(-2.0),\n"
        "                  np.float64(rng.normal()), np.float64(rng.normal())), 4000.0, float(0.3 * rng.normal()), True))\n"
        "serial = [hm._get_h(*j) for j in jobs]\n"
        "pooled = hm._get_h_many(jobs, workers=3).get()\n"
        "assert [tuple(map(float, a)) for a in serial] == [tuple(map(float, b)) for b in pooled]\n"
        "print('pool ok')\n" % (root, str(tmp_path)))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pool ok" in r.stdout, r.stdout + r.stderr
    assert len([f for f in os.listdir(tmp_path) if f.startswith("ran_")]) == 1  # the script body ran exactly once


def test_nearest_fft_number_api():
    import golden_util as gu
    from getdist_amd.convolve import nearestFFTnumber

    g = np.load(gu.GOLDEN_DIR + "/fftnumbers.npz")
    assert np.array_equal(nearestFFTnumber(g["x"]), g["y"])
