"""Host-side timeline of one bench step on the GPU box (GETDIST_AMD_HOSTLOG hooks in getdist_amd/mcsamples.py)."""
import os, sys, time
os.environ["GETDIST_AMD_HOSTLOG"] = "1"
sys.path.insert(0, ".")
import bench
from getdist_amd import mcsamples, synth
from getdist_amd.mcsamples import MCSamples
s, w, names, ranges = synth.config_c3()
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(len(names))
for _ in range(3):
    d = bench.one_step(mc, pairs, None, 0, 1, None)
mc.ctx.reserve_pinned_twin()
for _ in range(3):
    mcsamples._HOSTLOG.append((time.perf_counter(), "STEP START"))
    d = bench.one_step(mc, pairs, None, 0, 1, None)
    mcsamples._HOSTLOG.append((time.perf_counter(), "step returned"))
log = mcsamples._HOSTLOG
i0 = max(i for i, (t, wv) in enumerate(log) if wv == "STEP START")
t0 = log[i0][0]
for t, wv in log[i0:]:
    print("%8.3f ms  %s" % ((t - t0) * 1e3, wv))
