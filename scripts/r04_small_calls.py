"""gd_cov on an eighth of the rows and gd_quantiles_mm of seven columns (a rank's share of a step at 8 ranks), repeated:
for a kernel trace.  python scripts/r04_small_calls.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from getdist_amd import synth
from getdist_amd._lib import Context
s, w, names, ranges = synth.config_c3()
ctx = Context(0); ctx.upload(s, w)
N = ctx.N
mm = np.stack([s.min(0), s.max(0)], axis=1)
cols = np.arange(0, 50, 8)
t = np.tile(np.linspace(0.02, 0.98, 11) * N, (len(cols), 1))
for rep in range(12):
    t0 = time.perf_counter(); ctx.cov(None, 0, N // 8, minmax=True); t1 = time.perf_counter()
    ctx.quantiles(cols, t, 0, N, mm[cols]); t2 = time.perf_counter()
print("cov N/8: %.3f ms   quantiles 7 cols: %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
