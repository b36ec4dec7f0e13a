"""The real-weight 2D binning of a whole triangle alone (50 columns, 1225 pairs, N = 1e7, w ~ Exp(1)), for kernel traces /
counter passes.
python scripts/weighted_binning.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from getdist_amd import synth
from getdist_amd.mcsamples import MCSamples

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, n, F = 10_000_000, 50, 256
s2, w2, names2, ranges2 = synth.block_recipe(n, N, weighted=True, stream=4)
mc = MCSamples(samples=s2, weights=w2, names=names2, ranges=ranges2)
mc.prepareParams(neff=False)
e2 = [mc._bin_edges(p, F) for p in mc.paramNames.names]
idx = [mc._index_column(j, F, e2[j][1], e2[j][0]) for j in range(n)]
allp = synth.triangle_pairs(n)
out = mc.ctx.alloc(len(allp) * F * F * 8)
for _ in range(reps):
    mc.ctx.timer_start()
    mc.ctx.hist2d_prebinned([idx[a] for a, b in allp], [idx[b] for a, b in allp], F, out=out)
    print("weighted 2D binning of %d pairs: %.2f ms" % (len(allp), mc.ctx.timer_stop_ms()))
