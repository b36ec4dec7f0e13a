"""
Timings of the other BASELINE.json configs on one MI355X (not bench lines; recorded in DESIGN.md):
  C2: 1D KDE, 30 params, 1e7 weighted samples, mixed hard bounds
  C4: convergence, 8 chains x 5e6 x 100 params, weighted covariance + Gelman-Rubin (all chains on one GPU)
Usage: python scripts/run_configs.py [c2] [c4]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_c2():
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = synth.config_c2()
    t0 = time.perf_counter()
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    t_ctor = time.perf_counter() - t0
    mc.get1DDensities()
    times = []
    for _ in range(3):
        for p in mc.paramNames.names:
            p.N_eff_kde = None
            p._ranges_done = False
        mc._initLimits()
        mc.ctx.sync()
        t0 = time.perf_counter()
        d = mc.get1DDensities()
        mc.ctx.sync()
        times.append(time.perf_counter() - t0)
    t = min(times)
    return dict(config="C2", n=mc.n, N=mc.numrows, construct_s=round(t_ctor, 3), all_1d_densities_ms=round(t * 1e3, 2),
                densities_per_s=round(mc.n / t, 1), bounded=[p.name for p in mc.paramNames.names if p.has_limits],
                max_P=[float(x.P.max()) for x in d][:3])


def run_c4():
    import numpy as np

    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    s, w, names, offsets = synth.config_c4()
    chains = [s[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    ws = [w[a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    t0 = time.perf_counter()
    mc = MCSamples(samples=chains, weights=ws, names=names)
    t_ctor = time.perf_counter() - t0
    mc.getGelmanRubinEigenvalues()
    times = []
    for _ in range(3):
        mc._chain_stats_cache = {}  # (the per-chain moments are cached on the object: time their computation, not the look-up)
        mc.ctx.sync()
        t0 = time.perf_counter()
        D = mc.getGelmanRubinEigenvalues()
        mv = mc.getMeanVarTest()
        mc.ctx.sync()
        times.append(time.perf_counter() - t0)
    return dict(config="C4", chains=len(chains), N_per_chain=len(ws[0]), n=mc.n, construct_s=round(t_ctor, 3),
                gr_plus_meanvar_ms=round(min(times) * 1e3, 2), GR=float(np.max(D)), meanvar_max=float(np.max(mv)))


def run_c5(N=2_000_000, n=200):
    """Stress shape of C5 (200 parameters => 19 900 pairs + 200 1D densities + margestats) at a reduced row count."""
    import numpy as np

    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = synth.block_recipe(n, N, weighted=False, stream=7)
    t0 = time.perf_counter()
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    t_ctor = time.perf_counter() - t0
    t0 = time.perf_counter()
    ms = mc.getMargeStats()
    t_marge = time.perf_counter() - t0
    t0 = time.perf_counter()
    pairs, dens = mc.triangleDensities()
    dens[-1].P  # results are delivered lazily: the first read waits for this call's copies
    mc.ctx.sync()
    t_tri = time.perf_counter() - t0
    ok = all(d is not None and d.P.max() == 1.0 for d in dens)
    return dict(config="C5-reduced", n=n, N=N, pairs=len(pairs), construct_s=round(t_ctor, 2),
                margestats_1d_s=round(t_marge, 2), triangle_s=round(t_tri, 2), densities_per_s=round(len(pairs) / t_tri, 1),
                all_normalised=bool(ok), limits_p0=str(ms.parWithName("p0").limits[0]),
                F_classes=sorted(set(int(d.P.shape[0]) for d in dens)))


def run_latency(N=10_000_000, n=10):
    """Per-call latency of the drop-in API (what plots.py's per-pair loop would see)."""
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    s, w, names, ranges = synth.block_recipe(n, N, weighted=False, stream=31)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    pairs = synth.triangle_pairs(n)
    for a, b in pairs[:5]:
        mc.get2DDensity(a, b)
    t0 = time.perf_counter()
    for a, b in pairs[5:45]:
        mc.get2DDensity(a, b).P
    t2 = (time.perf_counter() - t0) / 40
    mc.get1DDensity(0)
    t0 = time.perf_counter()
    for j in range(1, n):
        mc.get1DDensity(j)
    t1 = (time.perf_counter() - t0) / (n - 1)
    t0 = time.perf_counter()
    mc.get2DDensities(pairs)[-1].P
    tb = (time.perf_counter() - t0) / len(pairs)
    return dict(config="latency", N=N, n=n, get2DDensity_ms=round(t2 * 1e3, 3), get1DDensity_ms=round(t1 * 1e3, 3),
                batched_2d_ms_per_pair=round(tb * 1e3, 3))


def main():
    which = sys.argv[1:] or ["c2", "c4"]
    for c in which:
        fn = {"c2": run_c2, "c4": run_c4, "c5": run_c5, "latency": run_latency}[c]
        print(json.dumps(fn()), flush=True)


if __name__ == "__main__":
    main()
