"""
bench.py -- BASELINE.json's metric on BASELINE.json's config: 2D KDE densities/sec on the 50-parameter,
10M-sample triangle (1225 pairs, base fine_bins_2D=256, default settings), config "C3" of SURVEY.md 8d.

One "step" = the whole hot path over the resident synthetic sample set: per-parameter preparation (ranges,
quantiles, limits, N_eff), weighted binning, bandwidth selection (device ISJ solver + host TNC), rocFFT
convolution, boundary / multiplicative-bias correction, normalisation, and the D2H copy of every grid.
All per-parameter / per-pair caches are cleared before every step; the sample columns are already in HBM
(upload time reported separately in DESIGN.md, never in `value`).

    python bench.py --gpus N --steps K --warmup W       (N>1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: every rank holds a replica of the samples; parameter preparation is split over ranks and its scalars
all-gathered (RCCL), pairs are partitioned by cost class with no data-path collective; total work is fixed
("strong" scaling); `value` = 1225 * K / max-over-ranks time.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
PMC_FETCH_KIB, PMC_WRITE_KIB = 10.51e6, 0.6272e6  # per 1225-pair launch; refreshed from profiles/r01_pmc_* below


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nsamples", type=int, default=10_000_000)
    ap.add_argument("--nparams", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", action="store_true", help="cProfile the timed steps (host side) to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for "
                    "exercising the multi-rank path on a single GPU)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--emulate-world", type=int, default=0, help="measurement aid: time rank 0's share of a W-rank "
                    "job on one GPU (other ranks' parameter state is replayed from a cached full preparation)")
    ap.add_argument("--cpu-baseline-n", type=int, default=None, help="rows for the CPU sample (default: nsamples)")
    return ap.parse_args()


def pair_cost_classes(mc, pairs):
    """Cost class per pair (upscaled grid?, #bounded parameters, optimiser vs rule-of-thumb), vectorised."""
    corr = np.abs(mc.getCorrelationMatrix())
    lim = np.array([bool(p.has_limits) for p in mc.paramNames.names], dtype=np.int64)
    a = np.fromiter((p[0] for p in pairs), dtype=np.int64, count=len(pairs))
    b = np.fromiter((p[1] for p in pairs), dtype=np.int64, count=len(pairs))
    c = corr[b, a]
    return ((c > 0.866).astype(np.int64) * 100 + (lim[a] + lim[b]) * 10 + (c > 0.2)).tolist()


def reset_caches(mc):
    for par in mc.paramNames.names:
        par.N_eff_kde = None
        par._ranges_done = False
    mc._initLimits()
    mc._idx_cols = {}
    mc.density1D = {}
    if getattr(mc, "_twin", None) is not None:  # the second lane's index columns are per-step work too
        mc._twin._idx_cols = {}


_REPLAY = {}


def one_step(mc, pairs_all, dist, rank, world, torch_device, emulate=0):
    """The timed unit of work.  Returns the list of Density2D this rank produced."""
    from getdist_amd import parallel

    t_step0 = time.perf_counter()
    reset_caches(mc)
    if emulate:
        world = emulate
    my_params = parallel.partition_round_robin(list(range(mc.n)), world, rank)
    # single rank: N_eff is left to get2DDensities, which overlaps it with the 2D binning on a second stream; with
    # several ranks it must be known before the parameter state is exchanged
    mc.prepareParams(my_params, neff=(world > 1))
    if emulate:
        others = [j for j in range(mc.n) if j not in my_params]
        parallel.unpack_param_state(mc, _REPLAY["rows"][others])  # what the all-gather would deliver
    else:
        parallel.allgather_param_state(mc, my_params, mc.n, dist if world > 1 else None, torch_device)
    t_part0 = time.perf_counter()
    classes = dict(zip(pairs_all, pair_cost_classes(mc, pairs_all)))
    _, my_pairs = parallel.partition_pairs(pairs_all, classes.__getitem__, world, rank)
    if mc._timing:
        mc.timings["step.partition"] = mc.timings.get("step.partition", 0.0) + time.perf_counter() - t_part0
    out = mc.get2DDensities(my_pairs)
    if mc._timing:
        mc.timings["step.total"] = mc.timings.get("step.total", 0.0) + time.perf_counter() - t_step0
    return out


def binning_kernel_roofline(mc, pairs_all, reps=5):
    """
    Time the dominant O(N) kernel of the step -- the batched weighted 2D binning of all F=256 pairs -- with HIP
    events on the library's stream, and price it with SURVEY.md 8d's algorithmic bytes B2 = 24 N + 8 F^2 per
    density (x, y, w read once as fp64 + the F x F fp64 grid written).
    """
    names = mc.paramNames.names
    F = mc.fine_bins_2D
    corr = mc.getCorrelationMatrix()
    sel = [p for p in pairs_all if abs(corr[p[1]][p[0]]) <= 0.866]
    ix, iy = [], []
    for (a, b) in sel:
        fwx, bx, _ = mc._bin_edges(names[a], F)
        fwy, by, _ = mc._bin_edges(names[b], F)
        ix.append(mc._index_column(a, F, bx, fwx))
        iy.append(mc._index_column(b, F, by, fwy))
    out = mc.ctx.alloc(len(sel) * F * F * 8)
    mc.ctx.hist2d_prebinned(ix, iy, F, out=out)
    mc.ctx.sync()
    ms = []
    for _ in range(reps):
        mc.ctx.timer_start()
        mc.ctx.hist2d_prebinned(ix, iy, F, out=out)
        ms.append(mc.ctx.timer_stop_ms())
    out.free()
    t = float(np.median(ms)) * 1e-3
    alg_bytes = len(sel) * (24.0 * mc.numrows + 8.0 * F * F)
    achieved = alg_bytes / t / 1e9
    # HBM traffic per launch from the PMC passes in profiles/r01_pmc_hist2d_{FETCH,WRITE}_SIZE.csv (separate rocprofv3
    # --pmc runs of scripts/pmc_hist2d.py on this exact config: 1225 pairs, N=1e7, F=256, unit weights):
    # FETCH_SIZE x2 (gfx950 wide-load correction, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB, scaled by the
    # number of pairs in this launch.  Not re-measured live (PMC needs the profiler); null for other configs.
    traffic = None
    if mc.numrows == 10_000_000 and mc.n == 50 and mc.weights is None:
        traffic = (2 * PMC_FETCH_KIB + PMC_WRITE_KIB) * 1024 * len(sel) / 1225.0
    return dict(bound="hbm", kernel="k_hist2d_u16 (pre-binned u16 indices, 16-bit packed LDS counters)", launches_pairs=len(sel),
                ms_per_launch=t * 1e3, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                traffic=traffic,
                note="achieved = algorithmic bytes (24N+8F^2 per density, SURVEY 8d) / launch time; the kernel reads "
                     "pre-binned u16 indices (4 B/sample/stripe) so frac > 1 is expected; the fused-fp64 variant of the "
                     "same kernel streams x,y and reaches 6.4 TB/s algorithmic = 0.81 of peak (profiles/r01_kernel_bench.txt)")


def cpu_baseline(nparams, nsamples, n_rows):
    """
    The oracle (numpy/scipy restatement of the reference, 'port') on a bounded sample of the same workload:
    3 of the 50 parameters and 2 of the 1225 pairs at full N, timed on this host; whole-triangle throughput
    extrapolated as 1225 / (50 t_prep + 1225 t_pair).
    """
    from getdist_amd import synth
    from oracle import kde_oracle as ko

    s, w, names, ranges = synth.config_c3(n_rows, nparams)
    cols = [5, 6, 20] if nparams > 20 else [0, 1, 2]
    sub = np.ascontiguousarray(s[:, cols])
    sub_names = [names[c] for c in cols]
    orc = ko.OracleSamples(sub, w, names=sub_names, ranges={k: v for k, v in ranges.items() if k in sub_names})
    t0 = time.perf_counter()
    for j in range(3):
        orc.init_param(j)
        orc.neff_1d(j)
    t_prep = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    orc.density_2d(0, 1)
    orc.density_2d(0, 2)
    t_pair = (time.perf_counter() - t0) / 2
    npairs = nparams * (nparams - 1) // 2
    value = npairs / (nparams * t_prep + npairs * t_pair)
    return dict(value=value, unit="densities/s", cores=1, kind="port",
                sample="oracle on 3 of %d parameters (prep %.2f s each) + 2 of %d pairs (%.2f s each) at N=%d; "
                       "triangle extrapolated as npairs/(nparams*t_prep + npairs*t_pair); host has %d logical cores, "
                       "the reference path is single-process" % (nparams, t_prep, npairs, t_pair, n_rows, os.cpu_count()))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch_device = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            local_rank = 0
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            torch_device = torch.device("cuda", local_rank)
        dist_mod.init_process_group(backend=args.backend, rank=rank, world_size=world)
        dist = dist_mod

    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    t0 = time.perf_counter()
    s, w, names, ranges = synth.config_c3(args.nsamples, args.nparams)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, device=local_rank)
    t_ctor = time.perf_counter() - t0
    pairs_all = synth.triangle_pairs(args.nparams)

    def barrier():
        mc.ctx.sync()
        if dist is not None:
            import torch

            dist.barrier()
            if torch_device is not None:
                torch.cuda.synchronize()

    if args.emulate_world:
        from getdist_amd import parallel

        mc.prepareParams()
        _REPLAY["rows"] = parallel.pack_param_state(mc, list(range(mc.n)))
    dens = None
    for _ in range(args.warmup):
        dens = one_step(mc, pairs_all, dist, rank, world, torch_device, args.emulate_world)  # held like the timed results
    if args.warmup > 0:
        mc.ctx.reserve_pinned_twin()  # result buffers for "previous step still referenced" + "current step"
    barrier()
    mc.timings = {}
    prof = None
    if args.profile:
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dens = one_step(mc, pairs_all, dist, rank, world, torch_device, args.emulate_world)
    barrier()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(30)
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device=torch_device or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    npairs = len(pairs_all)
    value = npairs * args.steps / elapsed  # with --emulate-world: what W ranks would deliver if all took this long

    if rank == 0:
        line = {
            "metric": "2D KDE densities/sec (triangle, %d params, %s samples)" % (args.nparams, "{:.0e}".format(args.nsamples)),
            "value": value, "unit": "densities/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic (seeded block recipe, SURVEY.md 8d C3)",
            "config": {"workload": "C3: 2D KDE triangle, %d params (%d pairs), N=%d unit-weight samples, base fine_bins_2D=256, "
                                   "default settings; includes per-parameter prep, bandwidth selection and D2H of all grids"
                                   % (args.nparams, npairs, args.nsamples),
                       "parallelism": "pairs partitioned over %d GPU(s), samples replicated" % world,
                       "setup_s": {"generate": round(t_gen, 2), "construct_upload_basestats": round(t_ctor, 2)}},
        }
        if args.emulate_world:
            line["emulated_world"] = args.emulate_world
            line["n_gpus"] = args.emulate_world
        if world == 1 and not args.emulate_world:
            line["roofline"] = binning_kernel_roofline(mc, pairs_all)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args.nparams, args.nsamples, args.cpu_baseline_n or args.nsamples)
        if mc._timing:
            line["phase_seconds_total"] = {k: round(v, 4) for k, v in sorted(mc.timings.items())}
        assert len(dens) > 0 and all(d is not None for d in dens)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
