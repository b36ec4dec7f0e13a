"""
Rank 0's share of a W-rank job timed on ONE GPU, for several W in one process (the sample set is generated and uploaded
once): the scaling evidence a 1-GPU box can give for the batched-pairs path.  The other ranks' contributions
to the step's exchanges are replayed from a full single-rank preparation, exactly as `bench.py --emulate-world W` does
(this script drives bench.one_step).

  python scripts/emulate_scaling.py --nparams 50  --nsamples 10000000 --worlds 1,2,4,8      # C3
  python scripts/emulate_scaling.py --nparams 200 --nsamples 2000000  --worlds 1,2,4,8      # C5's 19 900 pairs, reduced rows

Prints one JSON line: per W the ms per step of rank 0's share, the pairs and the columns it touched, and the efficiency
t(1) / (W t(W)).
"""
import argparse
import gc
import json
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nparams", type=int, default=50)
    ap.add_argument("--nsamples", type=int, default=10_000_000)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    args = ap.parse_args()
    import bench
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    logging.getLogger().setLevel(logging.ERROR)
    s, w, names, ranges = synth.config_c3(args.nsamples, args.nparams)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    pairs_all = synth.triangle_pairs(args.nparams)
    rows = []
    t1 = None
    for W in [int(x) for x in args.worlds.split(",")]:
        emu = 0 if W == 1 else W
        if emu:
            bench.prepare_replay(mc, W)
        for _ in range(args.warmup):
            dens = bench.one_step(mc, pairs_all, None, 0, 1, None, emu)
        mc.ctx.reserve_pinned_twin()
        if getattr(mc, "_twin", None) is not None:
            mc._twin.ctx.reserve_pinned_twin()
        gc.collect()
        gc.disable()
        mc.ctx.sync()
        mc.ctx.copy_sync()
        t0 = time.perf_counter()
        returned = [t0]
        for _ in range(args.steps):
            dens = bench.one_step(mc, pairs_all, None, 0, 1, None, emu)
            returned.append(time.perf_counter())
        if len(dens):
            dens[-1].P
        mc.ctx.sync()
        mc.ctx.copy_sync()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        between = np.diff(returned) * 1e3  # host time between the returns of consecutive steps (steady state = the GPU's pace)
        gc.enable()
        mine = np.asarray(bench._REPLAY["last_pairs"]).reshape(-1, 2)
        if W == 1:
            t1 = ms
        rows.append(dict(world=W, ms_per_step_rank0=round(ms, 3), ms_between_step_returns_median=round(float(np.median(between)), 3),
                         ms_between_step_returns=[round(float(x), 2) for x in between],
                         pairs_rank0=int(len(mine)), columns_touched_rank0=int(len(np.unique(mine))),
                         efficiency=None if t1 is None else round(t1 / (W * ms), 3)))
        print("W=%d  %.2f ms  pairs %d  columns %d" % (W, ms, len(mine), len(np.unique(mine))), file=sys.stderr)
    print(json.dumps(dict(what="rank 0's share of a W-rank step on one GPU (other ranks' exchange contributions replayed)",
                          nparams=args.nparams, nsamples=args.nsamples, pairs=len(pairs_all), steps=args.steps,
                          pair_deal=os.environ.get("GETDIST_AMD_PAIR_DEAL", "blocks"), rows=rows)))


if __name__ == "__main__":
    main()
