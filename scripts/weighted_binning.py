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

# round 6: the same histograms over BYTE indices, the samples partitioned by 16-row stripe once per y column (k_wpart_* +
# k_hist2d_wsorted, csrc/binning.hip): the route gd_density2d_batch takes for real weights
b8 = [mc.ctx.alloc(N + 64) for _ in range(n)]
mc.ctx.prebin8_batch(list(range(n)), [x[1] for x in e2], [x[0] for x in e2], 256, b8)
for _ in range(reps):
    mc.ctx.timer_start()
    mc.ctx.hist2d_prebinned8([b8[a] for a, b in allp], [b8[b] for a, b in allp], out=out)
    print("weighted 2D binning of %d pairs, byte indices sorted by stripe: %.2f ms" % (len(allp), mc.ctx.timer_stop_ms()))
