"""
Per-kernel timings on the MI355X through the C ABI, with HIP events on the library stream (gd_timer_*), priced in
bytes moved per second.  Used for DESIGN.md's kernel table; run on the GPU box:

    python scripts/kernel_bench.py [--nsamples 10000000] [--nparams 50]
"""

import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from getdist_amd import synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402


def timed(ctx, fn, reps=5):
    fn()
    ctx.sync()
    ms = []
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ms.append(ctx.timer_stop_ms())
    return float(np.median(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsamples", type=int, default=10_000_000)
    ap.add_argument("--nparams", type=int, default=50)
    ap.add_argument("--weighted", action="store_true")
    ap.add_argument("--intweights", action="store_true", help="integer multiplicities 1..5 as weights")
    a = ap.parse_args()
    s, w, names, ranges = synth.block_recipe(a.nparams, a.nsamples, weighted=a.weighted, stream=4)
    if a.intweights:
        w = np.random.default_rng(1).integers(1, 6, a.nsamples).astype(np.float64)
        a.weighted = True
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    ctx, N, n = mc.ctx, mc.numrows, mc.n
    wb = 8 if a.weighted else 0
    res = {}

    def rec(name, ms, nbytes, note=""):
        res[name] = dict(ms=round(ms, 4), GBps=round(nbytes / ms / 1e6, 1), bytes=int(nbytes), note=note)
        print("%-28s %9.3f ms  %9.1f GB/s  %s" % (name, ms, nbytes / ms / 1e6, note), flush=True)

    rec("col_stats(all cols)", timed(ctx, ctx.col_stats), 2 * n * N * (8 + wb), "2 passes x (x[,w])")
    rec("cov(all cols)", timed(ctx, ctx.cov, 3), n * N * 8 * 2 + 2 * N * wb, "means pass + SYRK pass; %.1f GFLOP" % (2 * N * 64 * 64 * ((n + 63) // 64) ** 2 / 1e9))
    fr = np.array([0.001, 0.999] + list(np.linspace(0.1, 0.9, 9)))
    tg = np.tile(mc.norm * fr, (n, 1))
    rec("quantiles(all cols, 11 q)", timed(ctx, lambda: ctx.quantiles(list(range(n)), tg), 3), 8 * n * N * (8 + wb), "8 radix passes")
    rec("autocov 32 lags (1 col)", timed(ctx, lambda: ctx.autocov_lags(3, mc.means[3], 0, 32)), N * (8 + wb) * 2, "")
    rec("kde lag sums 7 lags (1 col)", timed(ctx, lambda: ctx.kde_lag_sums(3, 1.0, [N // 2 + k for k in range(5)] + [1, 2])), 7 * N * 2 * (8 + wb), "")
    mc.prepareParams(neff=False)
    par = mc.paramNames.names
    e1 = [mc._bin_edges(p, 1024) for p in par]
    rec("hist1d(all cols, F=1024)", timed(ctx, lambda: ctx.hist1d(list(range(n)), [e[1] for e in e1], [e[0] for e in e1], 1024)), n * N * (8 + wb), "")
    F = 256
    e2 = [mc._bin_edges(p, F) for p in par]
    buf = ctx.alloc(N * 2 + 64)
    rec("prebin u16 (1 col)", timed(ctx, lambda: ctx.prebin(0, e2[0][1], e2[0][0], F, buf)), N * 10, "8 B in, 2 B out")
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    B2 = 24.0 * N + 8.0 * F * F
    idx = [mc._index_column(j, F, e2[j][1], e2[j][0]) for j in range(n)]
    out = ctx.alloc(len(pairs) * F * F * 8)
    for nb in (1, 64, len(pairs)):
        pp = pairs[:nb]
        ms = timed(ctx, lambda: ctx.hist2d_prebinned([idx[p[0]] for p in pp], [idx[p[1]] for p in pp], F, out=out), 3)
        rec("hist2d prebinned B=%d" % nb, ms, nb * B2, "algorithmic 24N+8F^2 per pair; real reads %d B/sample/stripe" % (4 + wb))
        ms = timed(ctx, lambda: ctx.hist2d([p[0] for p in pp], [p[1] for p in pp], [e2[p[0]][1] for p in pp], [e2[p[0]][0] for p in pp],
                                           [e2[p[1]][1] for p in pp], [e2[p[1]][0] for p in pp], F, out=out), 3)
        rec("hist2d direct fp64 B=%d" % nb, ms, nb * B2, "algorithmic 24N+8F^2 per pair")
    # 2D bandwidth optimiser on the 1225 histograms (half with the odd functionals)
    hp = ctx.hist2d_prebinned([idx[p[0]] for p in pairs], [idx[p[1]] for p in pairs], F, out=out)
    neff = [float(N)] * len(pairs)
    for dc in (0, 1):
        ms = timed(ctx, lambda: ctx.kopt2d(hp, len(pairs), F, neff, [dc] * len(pairs), [1e-4] * len(pairs), [0.3] * len(pairs)), 3)
        rec("kopt2d B=%d do_corr=%d" % (len(pairs), dc), ms, len(pairs) * 8.0 * F * F, "DCT GEMMs + device Brent + psi functionals + get_h (TNC)")
    info = ctx.device_info()
    print(json.dumps(dict(N=N, n=n, weighted=a.weighted, device=info, kernels=res)))


if __name__ == "__main__":
    main()
