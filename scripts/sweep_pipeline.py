"""Step time of C3 on one GPU against the host-side pipeline knobs (class attributes of MCSamples): the fraction of the
base-grid pairs in the first optimiser launch and the pair count from which the launch is split at all.  One data set,
one process; per setting 5 warm-up + 30 timed steps, median of the step-to-step period.
    python scripts/sweep_pipeline.py  ->  gpurun_out/r03_sweep_pipeline.json"""
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from getdist_amd import synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402

s, w, names, ranges = synth.config_c3()
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(len(names))
for _ in range(4):
    d = bench.one_step(mc, pairs, None, 0, 1, None)
mc.ctx.reserve_pinned_twin()
if mc._twin is not None:
    mc._twin.ctx.reserve_pinned_twin()
gc.collect(); gc.freeze(); gc.disable()
res = []
settings = [dict(KOPT_FIRST_FRACTION=f, KOPT_SPLIT_MIN=256) for f in (0.5, 0.35, 0.42, 0.58, 0.66)]
settings += [dict(KOPT_FIRST_FRACTION=0.5, KOPT_SPLIT_MIN=10 ** 9), dict(KOPT_FIRST_FRACTION=0.5, KOPT_SPLIT_MIN=256)]
for st in settings:
    for k, v in st.items():
        setattr(MCSamples, k, v)
    for _ in range(5):
        d = bench.one_step(mc, pairs, None, 0, 1, None)
    ts = []
    t0 = time.perf_counter()
    for _ in range(30):
        d = bench.one_step(mc, pairs, None, 0, 1, None)
        t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e3)
        t0 = t1
    d[-1].P  # the last step's grids have landed
    r = dict(st, ms_median=float(np.median(ts)), ms_mean=float(np.mean(ts)), ms_min=float(np.min(ts)))
    print(r, flush=True)
    res.append(r)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "r03_sweep_pipeline.json"), "w"), indent=1)
