import sys, time
sys.path.insert(0, ".")
import numpy as np
from getdist_amd import synth
from getdist_amd.mcsamples import MCSamples
s2, w2, names2, ranges2 = synth.config_c3(1_000_000, 50)
pairs = synth.triangle_pairs(50)
mc2 = MCSamples(samples=s2, weights=w2, names=names2, ranges=ranges2)
for rep in range(4):
    for p in mc2.paramNames.names:
        p.N_eff_kde = None
        p._ranges_done = False
    mc2.ctx.sync()
    t0 = time.perf_counter()
    mc2.updateBaseStatistics()
    d2 = mc2.get2DDensities(pairs)
    t1 = time.perf_counter()
    d2[-1].P
    mc2.ctx.sync(); mc2.ctx.copy_sync()
    t2 = time.perf_counter()
    print("rep %d: returned after %.1f ms, complete after %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
