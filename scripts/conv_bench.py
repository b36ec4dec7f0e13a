"""The convolution stage alone on one batch of 136 random F = 256 histograms (S = 288 frames, bounded and unbounded pairs,
linear boundary correction + one bias-correction round): for kernel traces of the LDS-transform kernels.
python scripts/conv_bench.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from getdist_amd._lib import Context

F, B = 256, 136
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
r = np.random.default_rng(9)
yy, xx = np.mgrid[0:F, 0:F]
hists = np.empty((B, F, F))
for b in range(B):
    cx, cy = r.uniform(0.3, 0.7, 2) * F
    sx, sy = r.uniform(0.05, 0.15, 2) * F
    hists[b] = r.poisson(3000.0 * np.exp(-0.5 * (((xx - cx) / sx) ** 2 + ((yy - cy) / sy) ** 2))).astype(np.float64)
rx, ry = r.uniform(2.0, 3.2, B), r.uniform(2.0, 3.2, B)
corr = r.uniform(-0.6, 0.6, B)
winw = np.maximum(1, np.rint(2.5 * np.maximum(rx, ry))).astype(np.int32)
flags = np.where(np.arange(B) % 3 == 0, 5 | 64, 0).astype(np.int32)
ctx = Context(0)
ctx.upload(r.standard_normal((1000, 2)), None)
d_hist = ctx.alloc(hists.nbytes)
d_hist.from_host(hists)
ts = []
chk = None
for rep in range(reps):
    t0 = time.perf_counter()
    d_P, status = ctx.density2d(d_hist, B, F, rx, ry, corr, winw, flags, 1, 1)
    ts.append(time.perf_counter() - t0)
    if rep == 0:
        chk = d_P.to_host((B, F, F)).copy()
    d_P.free()
import zlib
print("density2d %d pairs S=288: %.3f ms (min of %d); grid checksum %08x sum %.12f" % (B, 1e3 * min(ts[1:]), reps - 1, zlib.crc32(chk.tobytes()), float(chk.sum())))
