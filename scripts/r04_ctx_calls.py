"""Every call through the C ABI during the statistics + preparation phases of one emulated-rank step, with its host wall
time (the calls are blocking there): python scripts/r04_ctx_calls.py [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from getdist_amd import synth, parallel
from getdist_amd._lib import Context
from getdist_amd.mcsamples import MCSamples
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s, w, names, ranges = synth.config_c3()
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(len(names))
bench.prepare_replay(mc, W)
for _ in range(4):
    bench.one_step(mc, pairs, None, 0, 1, None, W)
LOG = []
def wrap(name, fn):
    def inner(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        LOG.append((name, (time.perf_counter() - t0) * 1e3, t0))
        return r
    return inner
for name in dir(Context):
    if name.startswith("_"): continue
    fn = getattr(Context, name)
    if callable(fn) and name not in ("close",):
        setattr(Context, name, wrap(name, fn))
t00 = time.perf_counter()
mc.updateBaseStatistics(row_share=(0, W), exchange=lambda mine: [mine] + bench._REPLAY["moments"][W][1:])
t1 = time.perf_counter()
bench.reset_caches(mc)
my_params = parallel.partition_round_robin(list(range(mc.n)), W, 0)
mc.prepareParams(my_params, neff=False)
t2 = time.perf_counter()
print("updateBaseStatistics %.3f ms, prepareParams %.3f ms" % ((t1 - t00) * 1e3, (t2 - t1) * 1e3))
for name, ms, t0 in LOG:
    print("  +%7.3f  %-28s %7.3f ms" % ((t0 - t00) * 1e3, name, ms))
