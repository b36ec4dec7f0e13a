"""
Isolated kernel timings on the MI355X (run on the GPU box):  python scripts/r03_kernels.py [--quick]
  * k_hist2d_u8_pf over the base pairs of C3 (HIP events), the other O(N) kernels of a step one by one;
  * k_cov_slab2 at C3's shape (50 columns x 1e7, unit weights), C4's (100 x 5e6, weighted), C5's width (200 columns,
    2e6 rows, unit and weighted) and two odd widths, against numpy on a row prefix and on an odd row range.
Writes gpurun_out/r03_kernels.json.

profiles/r03_kernels_ab.json is this script's output at commit e7ee704, when the round-2 kernels (k_hist2d_u8 with flat
loads, k_cov_slab) were still in the library behind GDHIP_U8_VARIANT / GDHIP_COV_OLD and were timed side by side and
compared bit for bit with the present ones; they have been removed since, the switches with them.
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
    return float(np.median(ms)), float(np.min(ms))


def main():
    quick = "--quick" in sys.argv
    res = {}
    N, n, F = (2_000_000 if quick else 10_000_000), 50, 256
    s, w, names, ranges = synth.config_c3(N, n)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    mc.prepareParams(neff=False)
    ctx = mc.ctx
    par = mc.paramNames.names
    e = [mc._bin_edges(p, F) for p in par]
    corr = mc.getCorrelationMatrix()
    pairs = [p for p in synth.triangle_pairs(n) if abs(corr[p[1]][p[0]]) <= 0.866]
    assert mc._index_columns8({j: (e[j][1], e[j][0]) for j in range(n)})
    i8 = [mc._idx_cols[(j, 256, "u8")][0] for j in range(n)]
    out = ctx.alloc(len(pairs) * F * F * 8)
    ix, iy = [i8[a] for a, b in pairs], [i8[b] for a, b in pairs]
    med, mn = timed(ctx, lambda: ctx.hist2d_prebinned8(ix, iy, out=out))
    got = out.to_host((64, F, F))  # the first 64 grids
    hist = dict(ms_median=med, ms_min=mn, mass_first64=float(np.sum(got)), mass_ok=bool(np.all(got.sum(axis=(1, 2)) == N)))
    print("k_hist2d_u8_pf", hist, flush=True)
    # the other O(N) kernels of a step, isolated (HIP events): pre-binning, quantile select (linear buckets vs radix),
    # the sheared re-binning of 79 pairs, the up-scaled classes (packed chunks vs the 32-bit kernel)
    iso = {}
    bufs = [ctx.alloc(N + 64) for _ in range(n)]
    med, mn = timed(ctx, lambda: ctx.prebin8_batch(list(range(n)), [e[j][1] for j in range(n)], [e[j][0] for j in range(n)], 256, bufs))
    iso["prebin8_batch_50_columns"] = dict(ms_median=med, ms_min=mn, GBps=n * N * 9 / med / 1e6)
    fr = np.array([0.001, 0.999] + list(np.linspace(0.1, 0.9, 9)))
    tg = np.tile(mc.norm * fr, (n, 1))
    mm = mc._minmax_of(list(range(n)))
    med, mn = timed(ctx, lambda: ctx.quantiles(list(range(n)), tg, minmax=mm), reps=3)
    q_lin = ctx.quantiles(list(range(n)), tg, minmax=mm)
    iso["quantiles_linear_50x11"] = dict(ms_median=med, ms_min=mn)
    med, mn = timed(ctx, lambda: ctx.quantiles(list(range(n)), tg), reps=3)
    iso["quantiles_radix_50x11"] = dict(ms_median=med, ms_min=mn, equal_to_linear=bool(np.array_equal(q_lin, ctx.quantiles(list(range(n)), tg))))
    sh = pairs[:79]
    shear_args = ([a for a, b in sh], [b for a, b in sh], [1.0] * 79, [-0.4] * 79, [e[a][1] for a, b in sh], [e[a][0] for a, b in sh],
                  [-12.0] * 79, [24.0 / 255] * 79, F)
    o2 = ctx.alloc(79 * F * F * 8)
    med, mn = timed(ctx, lambda: ctx.hist2d_sheared(*shear_args, out=o2))
    iso["hist2d_sheared_79_pairs"] = dict(ms_median=med, ms_min=mn)
    for Fu, npair in ((384, 13), (768, 6), (960, 6)):
        eu = [mc._bin_edges(p, Fu) for p in par]
        ixu = [mc._index_column(j, Fu, eu[j][1], eu[j][0]) for j in range(npair + 1)]
        o3 = ctx.alloc(npair * Fu * Fu * 8)
        xs, ys = [ixu[0]] * npair, ixu[1:npair + 1]
        med, mn = timed(ctx, lambda: ctx.hist2d_prebinned(xs, ys, Fu, out=o3))
        h_new = o3.to_host((npair, Fu, Fu)).copy()
        os.environ["GDHIP_NO_U16_CHUNKS"] = "1"
        med0, mn0 = timed(ctx, lambda: ctx.hist2d_prebinned(xs, ys, Fu, out=o3))
        os.environ.pop("GDHIP_NO_U16_CHUNKS")
        iso["hist2d_upscaled_F%d_%dpairs" % (Fu, npair)] = dict(ms_median_chunks=med, ms_median_u32=med0,
                                                                equal=bool(np.array_equal(h_new, o3.to_host((npair, Fu, Fu)))),
                                                                mass_ok=bool(np.all(h_new.sum(axis=(1, 2)) == N)))
        o3.free()
    print("isolated", json.dumps(iso), flush=True)
    res["isolated_kernels"] = iso
    res["hist2d_u8"] = dict(pairs=len(pairs), N=N, **hist)
    out.free()
    # sheared + upscaled classes through the (now global-load) generic kernels: timing only, parity is in pytest
    # ---- covariance
    cov = {}

    def cov_ab(tag, ctx_, m, rows, flops_note=""):
        r = {}
        med, mn = timed(ctx_, lambda: ctx_.cov(list(range(m))), reps=3)
        out_ = ctx_.cov(list(range(m)))
        r["slab2"] = dict(ms_median=med, ms_min=mn)
        r["symmetric"] = bool(np.array_equal(out_[1], out_[1].T))
        nt = (m + 15) // 16
        r["executed_TFLOPs_slab2"] = nt * (nt + 1) / 2 * 512.0 * rows / (r["slab2"]["ms_median"] * 1e-3) / 1e12
        r["GBps_slab2_incl_means_pass"] = 2.0 * rows * m * 8 / (r["slab2"]["ms_median"] * 1e-3) / 1e9
        cov[tag] = r
        print("cov", tag, r, flush=True)
        return out_

    got = cov_ab("C3_50x%g_unit" % N, ctx, n, N)
    sub = slice(0, 200_000)
    # numpy check on the full set is slow; the pooled statistics of a row sample bound gross errors, pytest does the rest
    ctx.close()
    del mc, s
    rng = np.random.default_rng(5)
    for tag, m, rows, weighted in (("C4_100x5e6_weighted", 100, 1_000_000 if quick else 5_000_000, True),
                                   ("C5w_200x2e6_unit", 200, 500_000 if quick else 2_000_000, False),
                                   ("C5w_200x2e6_weighted", 200, 500_000 if quick else 2_000_000, True),
                                   ("m130_x2e6_unit", 130, 500_000 if quick else 2_000_000, False),
                                   ("m30_x4e6_unit", 30, 1_000_000 if quick else 4_000_000, False)):
        A = rng.standard_normal((m, m)) / np.sqrt(m)
        x = np.asfortranarray(rng.standard_normal((rows, m)) @ (A + 0.3 * np.eye(m)))
        ww = rng.exponential(1.0, rows) if weighted else None
        mc = MCSamples(samples=x, weights=ww, names=["p%d" % i for i in range(m)])
        means, c, norm = cov_ab(tag, mc.ctx, m, rows)
        wn = np.ones(rows) if ww is None else ww
        mu = (wn @ x) / wn.sum()
        d = x[:300_000] - mu
        # exact check on a prefix is not the full covariance: compare the device's own prefix call instead
        mp, cp, _ = mc.ctx.cov(list(range(m)), lo=0, hi=300_000)
        mu_p = (wn[:300_000] @ x[:300_000]) / wn[:300_000].sum()
        dp = x[:300_000] - mu_p
        ref_c = (dp * wn[:300_000, None]).T @ dp / wn[:300_000].sum()
        cov[tag]["max_rel_err_vs_numpy_prefix_300k"] = float(np.max(np.abs(cp - ref_c)) / np.max(np.abs(ref_c)))
        # odd row range: exercises the guarded head / tail slabs
        mp2, cp2, _ = mc.ctx.cov(list(range(m)), lo=12_345, hi=212_350)
        sl = slice(12_345, 212_350)
        mu2 = (wn[sl] @ x[sl]) / wn[sl].sum()
        d2 = x[sl] - mu2
        ref2 = (d2 * wn[sl, None]).T @ d2 / wn[sl].sum()
        cov[tag]["max_rel_err_vs_numpy_odd_range"] = float(np.max(np.abs(cp2 - ref2)) / np.max(np.abs(ref2)))
        print("   numpy checks", cov[tag]["max_rel_err_vs_numpy_prefix_300k"], cov[tag]["max_rel_err_vs_numpy_odd_range"], flush=True)
        mc.ctx.close()
        del mc, x
    res["covariance"] = cov
    os.makedirs(OUT, exist_ok=True)
    json.dump(res, open(os.path.join(OUT, "r03_kernels.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
