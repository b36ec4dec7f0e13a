"""
Isolated kernel timings on the MI355X (HIP events through the C ABI):  python scripts/kernels_isolated.py
  the O(N) kernels of a C3 step one by one (statistics, quantile select, N_eff lag sums, pre-binning, byte-index binning,
  sheared min/max + re-binning), the optimiser in its two stages with both DCT routes, real-weight and integer-weight 2D
  binning of the whole triangle.  Writes gpurun_out/kernels_isolated.json.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from getdist_amd import synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def timed(ctx, fn, reps=5):
    fn()
    ctx.sync()
    ms = []
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ms.append(ctx.timer_stop_ms())
    return round(float(np.median(ms)), 4)


def main():
    res = {}
    N, n, F = 10_000_000, 50, 256
    s, w, names, ranges = synth.config_c3(N, n)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    mc.prepareParams(neff=False)
    ctx = mc.ctx
    par = mc.paramNames.names
    e = [mc._bin_edges(p, F) for p in par]
    corr = mc.getCorrelationMatrix()
    cols = list(range(n))
    res["base_statistics_cov_minmax_50"] = timed(ctx, lambda: ctx.cov(cols, minmax=True))
    fr = np.array([0.001, 0.999] + list(np.linspace(0.1, 0.9, 9)))
    tg = np.tile(mc.norm * fr, (n, 1))
    mm = mc._minmax_of(cols)
    res["quantiles_linear_50x11"] = timed(ctx, lambda: ctx.quantiles(cols, tg, minmax=mm), 3)
    res["autocov_probe_8_lags_50"] = timed(ctx, lambda: ctx.autocov_lags_batch(cols, mc.means, 0, 8))
    # round 6: the select's counting pass also writes the bucket columns and computes the lag probe (one read for the three)
    res["quantiles_linear_50x11_with_lag_probe_and_bucket_columns"] = timed(ctx, lambda: ctx.quantiles_probe(cols, tg, mm, mc.means), 3)
    os.environ["GDHIP_QLIN_UNFUSED"] = "1"
    res["quantiles_linear_50x11_unfused_round5_kernels"] = timed(ctx, lambda: ctx.quantiles(cols, tg, minmax=mm), 3)
    os.environ.pop("GDHIP_QLIN_UNFUSED")
    ctx.quantiles_probe(cols, tg, mm, mc.means)  # (bucket columns valid for what follows)
    kstd = np.array([(p.sigma_range or mc.sddev[j]) * 0.2 for j, p in enumerate(par)])
    lags = mc._neff_lag_list()
    res["kde_lag_sums_7_lags_50"] = timed(ctx, lambda: ctx.kde_lag_sums_batch(cols, 1.0 / (4 * kstd**2), lags), 3)
    bufs = [ctx.alloc(N + 64) for _ in range(n)]
    res["prebin8_batch_50_from_bucket_columns"] = timed(ctx, lambda: ctx.prebin8_batch(cols, [e[j][1] for j in cols], [e[j][0] for j in cols], 256, bufs))
    os.environ["GDHIP_NO_BUCKET_COLS"] = "1"
    res["prebin8_batch_50_from_fp64_samples"] = timed(ctx, lambda: ctx.prebin8_batch(cols, [e[j][1] for j in cols], [e[j][0] for j in cols], 256, bufs))
    os.environ.pop("GDHIP_NO_BUCKET_COLS")
    b16 = [ctx.alloc(2 * N + 64) for _ in range(12)]
    e960 = [mc._bin_edges(p, 960) for p in par[:12]]
    res["prebin_u16_12_columns_F960_from_bucket_columns"] = timed(ctx, lambda: ctx.prebin_batch(cols[:12], [x[1] for x in e960], [x[0] for x in e960], 960, b16))
    os.environ["GDHIP_NO_BUCKET_COLS"] = "1"
    res["prebin_u16_12_columns_F960_from_fp64_samples"] = timed(ctx, lambda: ctx.prebin_batch(cols[:12], [x[1] for x in e960], [x[0] for x in e960], 960, b16))
    os.environ.pop("GDHIP_NO_BUCKET_COLS")
    for b in b16:
        b.free()
    pairs = [p for p in synth.triangle_pairs(n) if abs(corr[p[1]][p[0]]) <= 0.866]
    out = ctx.alloc(len(pairs) * F * F * 8)
    ix, iy = [bufs[a] for a, b in pairs], [bufs[b] for a, b in pairs]
    res["hist2d_u8_%d_pairs" % len(pairs)] = timed(ctx, lambda: ctx.hist2d_prebinned8(ix, iy, out=out))
    A = [(a, b) for (a, b) in synth.triangle_pairs(n) if 0.2 < abs(corr[b][a]) <= mc.max_corr_2D and not (par[a].has_limits and par[b].has_limits)]
    ci, cj = [a for a, b in A], [b for a, b in A]
    r0 = np.full(len(A), -0.5)
    r1 = np.ones(len(A))
    res["minmax_affine_%d_sheared_pairs" % len(A)] = timed(ctx, lambda: ctx.minmax_affine(ci, cj, r0, r1))
    m2 = ctx.minmax_affine(ci, cj, r0, r1)
    xmin = np.array([mc._col_min[a] for a in ci]) - 0.1
    dx = (np.array([mc._col_max[a] for a in ci]) + 0.1 - xmin) / (F - 1)
    ymin = m2[:, 0] - 0.1
    dy = (m2[:, 1] + 0.1 - ymin) / (F - 1)
    d_rot = ctx.alloc(len(A) * F * F * 8)
    res["hist2d_sheared_%d_pairs" % len(A)] = timed(ctx, lambda: ctx.hist2d_sheared(ci, cj, r0, r1, xmin, dx, ymin, dy, F, out=d_rot))
    # the optimiser on the triangle's own histograms: whole call, both DCT routes
    B = len(pairs)
    neff = np.full(B, float(N))
    dc = np.array([0 if (par[a].has_limits or par[b].has_limits) else 1 for a, b in pairs], dtype=np.int32)
    fb = np.full(B, 1e-4)
    cc = np.array([corr[b][a] for a, b in pairs])
    for route in ("fft", "gemm"):
        if route == "gemm":
            os.environ["GDHIP_KOPT_DCT_GEMM"] = "1"
        else:
            os.environ.pop("GDHIP_KOPT_DCT_GEMM", None)
        res["kopt2d_whole_call_%d_pairs_dct_%s" % (B, route)] = timed(ctx, lambda: ctx.kopt2d(out, B, F, neff, dc, fb, cc), 3)
    os.environ.pop("GDHIP_KOPT_DCT_GEMM", None)
    # the fixed-point kernel alone: matrix resident on the CU (default) against streamed through LDS
    os.environ["GDHIP_KOPT_STREAMED"] = "1"
    res["kopt2d_whole_call_%d_pairs_streamed_fixed_point_kernel" % B] = timed(ctx, lambda: ctx.kopt2d(out, B, F, neff, dc, fb, cc), 3)
    os.environ.pop("GDHIP_KOPT_STREAMED", None)
    mc.ctx.close()
    # real weights and integer multiplicities: the whole triangle's 2D binning
    for tag, wts in (("real_weights", None), ("integer_weights", "int")):
        s2, w2, names2, ranges2 = synth.block_recipe(n, N, weighted=True, stream=4)
        if wts == "int":
            w2 = np.random.default_rng(1).integers(1, 6, N).astype(np.float64)
        mc2 = MCSamples(samples=s2, weights=w2, names=names2, ranges=ranges2)
        mc2.prepareParams(neff=False)
        e2 = [mc2._bin_edges(p, F) for p in mc2.paramNames.names]
        idx = [mc2._index_column(j, F, e2[j][1], e2[j][0]) for j in range(n)]
        allp = synth.triangle_pairs(n)
        o2 = mc2.ctx.alloc(len(allp) * F * F * 8)
        res["hist2d_prebinned_%s_%d_pairs" % (tag, len(allp))] = timed(
            mc2.ctx, lambda: mc2.ctx.hist2d_prebinned([idx[a] for a, b in allp], [idx[b] for a, b in allp], F, out=o2), 3)
        if wts is None:  # round 6: real weights over byte indices, samples partitioned by stripe once per y column
            b8 = [mc2.ctx.alloc(N + 64) for _ in range(n)]
            mc2.ctx.prebin8_batch(list(range(n)), [x[1] for x in e2], [x[0] for x in e2], 256, b8)
            res["hist2d_byte_index_real_weights_sorted_by_stripe_%d_pairs" % len(allp)] = timed(
                mc2.ctx, lambda: mc2.ctx.hist2d_prebinned8([b8[a] for a, b in allp], [b8[b] for a, b in allp], out=o2), 3)
        mc2.ctx.close()
    os.makedirs(OUT, exist_ok=True)
    json.dump(res, open(os.path.join(OUT, "kernels_isolated.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
