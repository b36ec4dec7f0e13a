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
# finer marks inside the Python part of the batched 2D call (module functions of getdist_amd/batch2d.py and a few methods)
from getdist_amd import batch2d as _b2  # noqa: E402


def _mark_fn(owner, name, label=None):
    fn = getattr(owner, name)

    def wrap(*a, _fn=fn, _name=label or name, **k):
        mcsamples._hostlog("   > " + _name)
        try:
            return _fn(*a, **k)
        finally:
            mcsamples._hostlog("   < " + _name)

    setattr(owner, name, wrap)


for _name in ("settings_of", "pack_params"):
    _mark_fn(_b2, _name)
for _name in ("getCorrelationMatrix", "getCov", "_init_params", "updateBaseStatistics"):
    _mark_fn(MCSamples, _name)
_mark_fn(type(mc.ctx), "pinned_array")
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
