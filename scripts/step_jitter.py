"""Which phase of a step is longer in the slow steps?  60 bench steps with the host log on; per phase: median and the values
in the steps that took more than 1.08 x the median step."""
import os, sys, time
os.environ["GETDIST_AMD_HOSTLOG"] = "1"
sys.path.insert(0, ".")
import gc
import numpy as np
import bench
from getdist_amd import mcsamples, synth
from getdist_amd.mcsamples import MCSamples
s, w, names, ranges = synth.config_c3()
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(len(names))
for _ in range(5):
    d = bench.one_step(mc, pairs, None, 0, 1, None)
mc.ctx.reserve_pinned_twin()
if mc._twin is not None:
    mc._twin.ctx.reserve_pinned_twin()
gc.collect(); gc.freeze(); gc.disable()
steps = []
for _ in range(60):
    mcsamples._HOSTLOG.clear()
    t0 = time.perf_counter()
    mcsamples._HOSTLOG.append((t0, "start"))
    d = bench.one_step(mc, pairs, None, 0, 1, None)
    mcsamples._HOSTLOG.append((time.perf_counter(), "returned"))
    steps.append(list(mcsamples._HOSTLOG))
mc.ctx.sync(); mc.ctx.copy_sync()
dur = np.array([st[-1][0] - st[0][0] for st in steps]) * 1e3
med = np.median(dur)
print("step ms: median %.2f  mean %.2f  max %.2f" % (med, dur.mean(), dur.max()))
labels = [l for _, l in steps[0]]
slow = [i for i, v in enumerate(dur) if v > 1.08 * med]
print("slow steps:", slow, [round(float(dur[i]), 1) for i in slow])
for k in range(1, len(labels)):
    seg = np.array([(st[k][0] - st[k - 1][0]) * 1e3 if len(st) == len(labels) else np.nan for st in steps])
    print("%-60s median %6.2f   slow: %s" % (labels[k][:60], np.nanmedian(seg), [round(float(seg[i]), 1) for i in slow]))
