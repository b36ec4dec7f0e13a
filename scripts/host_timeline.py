"""Python-side timeline of one bench step (GETDIST_AMD_HOSTLOG=1): where the interpreter spends the time between the blocking
C-ABI calls.    GETDIST_AMD_HOSTLOG=1 python scripts/host_timeline.py"""
import os
import sys
import time

os.environ["GETDIST_AMD_HOSTLOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from getdist_amd import mcsamples, synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402

s, w, names, ranges = synth.config_c3(10_000_000, 50)
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(50)
orig = {}
for name in ("cov", "quantiles_probe", "density2d_batch", "copy_wait", "batch2d_grid_sizes", "weight_stats"):
    fn = getattr(type(mc.ctx), name)

    def wrap(self, *a, _fn=fn, _name=name, **k):
        mcsamples._hostlog("-> " + _name)
        try:
            return _fn(self, *a, **k)
        finally:
            mcsamples._hostlog("<- " + _name)

    setattr(type(mc.ctx), name, wrap)
for _ in range(4):
    d = bench.one_step(mc, pairs, None, 0, 1, None)
    d[-1].P
del mcsamples._HOSTLOG[:]
t0 = time.perf_counter()
mcsamples._hostlog("step start")
d = bench.one_step(mc, pairs, None, 0, 1, None)
mcsamples._hostlog("one_step returned")
d[-1].P
mcsamples._hostlog("grids delivered")
prev = None
for t, what in mcsamples._HOSTLOG:
    print("%9.3f ms  (+%6.3f)  %s" % ((t - t0) * 1e3, 0.0 if prev is None else (t - prev) * 1e3, what))
    prev = t
