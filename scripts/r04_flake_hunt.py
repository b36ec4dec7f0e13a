"""Repeats the batched 2D calls of one fixture (alternating settings, as tests/test_gpu_densities.py does) and compares
every grid with the first result of the same settings: a grid that differs is a race.  Prints counts per route."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.fixtures import fixture_zoo
from getdist_amd.mcsamples import MCSamples

name = sys.argv[1] if len(sys.argv) > 1 else "periodic"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fx = {f["name"]: f for f in fixture_zoo()}[name]
out = {}
for route in os.environ.get("FLAKE_ROUTES", "native,python").split(","):
    os.environ["GETDIST_AMD_NATIVE_BATCH"] = "1" if route == "native" else "0"
    ref, bad = {}, []
    for rep in range(reps):
        mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"]) if rep % 10 == 0 else mc
        for ik, kw in enumerate(fx["kw2"]):
            if os.environ.get("FLAKE_KW") and str(ik) not in os.environ["FLAKE_KW"].split(","):
                continue
            pairs = fx["pairs"][::-1] if os.environ.get("FLAKE_REVERSE") else fx["pairs"]
            dens = mc.get2DDensities(pairs, get_density=False, **kw)
            for ip, d in enumerate(dens):
                P = np.array(d.P, copy=True)
                key = (ik, ip)
                if key not in ref:
                    ref[key] = P
                elif not np.array_equal(P, ref[key]):
                    bad.append((rep, ik, ip, float(np.max(np.abs(P - ref[key]))), tuple(d.bandwidth) if d.bandwidth else None))
    out[route] = dict(calls=reps * len(fx["kw2"]), bad=bad[:6], nbad=len(bad))
print(json.dumps(out))
