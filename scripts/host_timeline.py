"""Host-side timeline of one bench step on the GPU box (GETDIST_AMD_HOSTLOG hooks in getdist_amd/mcsamples.py).
   python scripts/host_timeline.py [W]     W > 1: rank 0's share of a W-rank job (as bench.py --emulate-world W)"""
import os, sys, time
os.environ["GETDIST_AMD_HOSTLOG"] = "1"
sys.path.insert(0, ".")
import bench
from getdist_amd import mcsamples, synth
from getdist_amd.mcsamples import MCSamples
W = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s, w, names, ranges = synth.config_c3()
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(len(names))
if W > 1:
    bench.prepare_replay(mc, W)
for _ in range(3):
    d = bench.one_step(mc, pairs, None, 0, 1, None, W)
mc.ctx.reserve_pinned_twin()
if mc._twin is not None:
    mc._twin.ctx.reserve_pinned_twin()
for _ in range(3):
    mcsamples._HOSTLOG.append((time.perf_counter(), "STEP START"))
    d = bench.one_step(mc, pairs, None, 0, 1, None, W)
    mcsamples._HOSTLOG.append((time.perf_counter(), "step returned"))
log = mcsamples._HOSTLOG
i0 = max(i for i, (t, wv) in enumerate(log) if wv == "STEP START")
t0 = log[i0][0]
for t, wv in log[i0:]:
    print("%8.3f ms  %s" % ((t - t0) * 1e3, wv))
