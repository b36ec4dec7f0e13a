import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from getdist_amd import synth
from getdist_amd.mcsamples import MCSamples
n, N = 200, 50_000_000
s, w, names, ranges = synth.block_recipe(n, N, weighted=False, stream=7)
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(n)
for route in (sys.argv[1],):
    os.environ["GETDIST_AMD_NATIVE_BATCH"] = route
    for p in mc.paramNames.names:
        p.N_eff_kde = None; p._ranges_done = False
    mc._initLimits()
    if hasattr(mc.ctx, "batch2d_invalidate"): mc.ctx.batch2d_invalidate()
    mc._idx_cols = {k: (buf, None) for k, (buf, _) in mc._idx_cols.items()}
    t0 = time.perf_counter(); ts = []
    for a in range(0, len(pairs), 2000):
        t1 = time.perf_counter()
        dens = mc.get2DDensities(pairs[a:a + 2000])
        t2 = time.perf_counter()
        for d in dens:
            d.P.max()
        ts.append((round(t2 - t1, 3), round(time.perf_counter() - t2, 3)))
    print("route native=%s: triangle %.2f s; per chunk (call, read): %s" % (route, time.perf_counter() - t0, ts), flush=True)
