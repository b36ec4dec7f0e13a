"""cProfile of the host side of one steady-state bench step (run on the GPU box)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from getdist_amd import synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402

s, w, names, ranges = synth.config_c3(10_000_000, 50)
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
pairs = synth.triangle_pairs(50)
for _ in range(2):
    bench.one_step(mc, pairs, None, 0, 1, None)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    d = bench.one_step(mc, pairs, None, 0, 1, None)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
