"""One chaotic TNC pair (fixture shapes_intweights, s0/s1): device functionals against the oracle's, the device triple, and
the oracle ensemble at several perturbation sizes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import kde_oracle as ko
from oracle.fixtures import fixture_zoo
from getdist_amd.mcsamples import MCSamples

fx = {f["name"]: f for f in fixture_zoo()}[sys.argv[1] if len(sys.argv) > 1 else "shapes_intweights"]
mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
for mode in ("resident", "streamed"):
    if mode == "streamed":
        os.environ["GDHIP_KOPT_STREAMED"] = "1"
    else:
        os.environ.pop("GDHIP_KOPT_STREAMED", None)
    mc._weightsChanged() if hasattr(mc, "_weightsChanged") else None
    d = mc.get2DDensities([fx["pairs"][0]], get_density=False)[0]
    tr = {}
    a, b = fx["pairs"][0]
    orc.density_2d(a, b, trace=tr)
    psi = np.array([tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"]])
    dev = np.array(d.kopt[1:7])
    print(mode, "t*", d.kopt[0], tr["t_star"], "psi rel err", np.abs(dev - psi) / np.abs(psi))
    print("  device triple", d.kopt[8:11], "oracle", tr["hx"], tr["hy"], tr["c"])
    p = np.zeros((5, 5)); p[0, 4], p[4, 0], p[2, 2], p[0, 0], p[1, 3], p[3, 1] = psi
    print("  amise device", ko.amise_from_psi(np.array(d.kopt[8:11]), p, tr["opt_N"]), "oracle", ko.amise_from_psi(np.array([tr["hx"], tr["hy"], tr["c"]]), p, tr["opt_N"]))
for rel in (1e-15, 1e-14, 1e-13, 1e-12):
    ens = ko.get_h_ensemble(tuple(psi), tr["opt_N"], tr["opt_corr"], rel=rel)
    am = np.array([ko.amise_from_psi(r, p, tr["opt_N"]) for r in ens])
    print("rel", rel, "hx", ens[:, 0].min(), ens[:, 0].max(), "c", ens[:, 2].min(), ens[:, 2].max(), "amise range", (am.max() - am.min()) / am.min())
