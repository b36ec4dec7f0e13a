"""
bench.py -- BASELINE.json's metric on BASELINE.json's config: 2D KDE densities/sec on the 50-parameter,
10M-sample triangle (1225 pairs, base fine_bins_2D=256, default settings), config "C3" of SURVEY.md 8d.

One "step" = the whole hot path over the resident synthetic sample set: base statistics (means, variances, covariance),
per-parameter preparation (ranges, quantiles, limits, N_eff), weighted binning, bandwidth selection (2D Botev fixed
point, psi functionals and the TNC refinement, all on the device), rocFFT convolution, boundary / multiplicative-bias
correction, normalisation, and the D2H copy of every grid.  All per-parameter / per-pair caches are cleared before every
step; the sample columns are already in HBM (upload time reported separately, never in `value`).

    python bench.py --gpus N --steps K --warmup W       (N>1: launched by torch.distributed.run, one rank per GPU)

After the timed region rank 0 adds, on one GPU only (SURVEY.md 8d protocol):
  * roofline: the batched 2D binning kernel timed with HIP events on the library's stream, priced by its HBM counter
    traffic (profiles/r02_pmc_hist2d.json, separate rocprofv3 --pmc passes) and by the unit-weight streaming model;
  * cpu_baseline: the oracle (numpy/scipy restatement of the reference) on the GPU box's host cores -- all 50
    per-parameter preparations plus up to 3 pairs of every (bandwidth branch, #bounded, grid size) class at N = 1e7,
    timed in a pool of min(cores, 32) single-threaded workers and extrapolated over the class census, a short
    single-process sample of the same tasks, and the UN-extrapolated full triangle at N = 1e6 on CPU and GPU;
  * parity: the GPU grids of those stratified pairs against the oracle's at full size; on hosts with >= 64 cores EVERY pair of
    the timed triangle at full size (parity.full_size_census: a pool sized to a third of the container's memory, longest
    pairs first, behind a wall-clock budget), else every sheared pair in addition to the stratified sample.  Each pair
    above 1e-6 carries its verdict record: which criterion of the frozen oracle-ensemble rule admitted it, at which
    perturbation scale, how far the nearest ensemble member is, and whether the strict slack of 0.25 admits it too.

Multi-GPU: every rank ends up with a replica of the samples -- it uploads its own block of columns and receives the others
over xGMI (parallel.ColumnShare: gd_upload_shard + gd_comm_share_columns) when the library communicator is available;
parameter preparation is split over ranks and its scalars all-gathered (RCCL inside the C ABI), pairs are dealt by tiles of
the triangle (a rank pre-bins only the columns its pairs touch) with no data-path collective; total work is fixed
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_hist2d.json")
if not os.path.exists(PMC_FILE):
    PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_hist2d.json")
LDS_ROOF_FILE = os.path.join(ROOT, "profiles", "r03_lds_atomic_roof.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nsamples", type=int, default=10_000_000)
    ap.add_argument("--nparams", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", action="store_true", help="cProfile the timed steps (host side) to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for "
                    "exercising the multi-rank path on a single GPU)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--emulate-world", type=int, default=0, help="measurement aid: time rank 0's share of a W-rank "
                    "job on one GPU (other ranks' parameter state is replayed from a cached full preparation)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU-baseline pool size (default min(cores, 32))")
    ap.add_argument("--cpu-small-n", type=int, default=1_000_000, help="rows of the un-extrapolated CPU/GPU triangle")
    ap.add_argument("--cpu-budget-s", type=float, default=240.0, help="wall-clock cap of each CPU-baseline pool stage; "
                    "a stage that exceeds it is abandoned and reported as such (the GPU numbers are printed regardless)")
    ap.add_argument("--census-budget-s", type=float, default=300.0, help="wall-clock cap of the all-pairs full-size parity "
                    "census (hosts with >= 64 cores); abandoned and reported as such beyond it")
    ap.add_argument("--context-factory", default="", help="testing only: 'module:attr' of a Context stand-in (the CPU "
                    "suite runs the launcher and the multi-rank step with tests/fake_ctx.py); the product path never sets it")
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """
    `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-launch this very command line
    under torch.distributed.run with one rank per GPU and hand its exit code back.  Under a launcher (the driver's
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) this is a no-op; main() then checks that the
    communicator really has --gpus ranks.
    """
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or args.emulate_world:
        return
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


_PAIR_INDEX = {}


def _pair_index(pairs):
    """The pair list of a run never changes: its index arrays are made once."""
    if _PAIR_INDEX.get("of") is not pairs:
        arr = np.array(pairs, dtype=np.int64).reshape(-1, 2)
        _PAIR_INDEX.update(of=pairs, array=arr, a=np.ascontiguousarray(arr[:, 0]), b=np.ascontiguousarray(arr[:, 1]))


def pair_cost_classes(mc, pairs):
    """Cost class per pair (upscaled grid?, #bounded parameters, optimiser vs rule-of-thumb), vectorised."""
    corr = np.abs(mc.getCorrelationMatrix())
    lim = np.array([bool(p.has_limits) for p in mc.paramNames.names], dtype=np.int64)
    _pair_index(pairs)
    a, b = _PAIR_INDEX["a"], _PAIR_INDEX["b"]
    c = corr[b, a]
    return (c > 0.866).astype(np.int64) * 100 + (lim[a] + lim[b]) * 10 + (c > 0.2)


def reset_caches(mc):
    for par in mc.paramNames.names:
        par.N_eff_kde = None
        par._ranges_done = False
    mc._initLimits()
    # the index columns are per-step work: every entry is marked stale (its device block is kept for the rewrite)
    mc._idx_cols = {k: (buf, None) for k, (buf, _) in mc._idx_cols.items()}
    mc.density1D = {}
    if hasattr(mc.ctx, "batch2d_invalidate"):
        mc.ctx.batch2d_invalidate()  # (the library's cache of index columns: binned again every step)
    if getattr(mc, "_twin", None) is not None:  # the second lane's as well
        mc._twin._idx_cols = {k: (buf, None) for k, (buf, _) in mc._twin._idx_cols.items()}


_REPLAY = {}


def _hostlog(what):
    from getdist_amd import mcsamples

    mcsamples._hostlog(what)


def prepare_replay(mc, W):
    """--emulate-world: what the other W - 1 ranks would contribute to the two all-gathers of a step."""
    from getdist_amd import parallel

    mc.prepareParams()
    _REPLAY["rows"] = parallel.pack_param_state(mc, list(range(mc.n)))
    per = (mc.numrows + W - 1) // W
    _REPLAY["moments"] = {W: [mc._partial_moments(min(r * per, mc.numrows), min((r + 1) * per, mc.numrows)) for r in range(W)]}


def one_step(mc, pairs_all, dist, rank, world, torch_device, emulate=0, comm=None):
    """The timed unit of work.  Returns the list of Density2D this rank produced.  ``comm``: a parallel.LibraryComm (RCCL
    through the C ABI) carries the step's three small exchanges instead of torch.distributed."""
    from getdist_amd import parallel

    t_step0 = time.perf_counter()
    if emulate:
        world = emulate
    if world > 1:
        # rows split over the ranks for the base statistics; the shares are pooled by one small all-gather
        if emulate:
            mc.updateBaseStatistics(row_share=(rank, world), exchange=lambda mine: [mine] + _REPLAY["moments"][world][1:])
        else:
            mc.updateBaseStatistics(row_share=(rank, world), exchange=lambda mine: parallel.allgather_vector(mine, dist, torch_device, comm))
    else:
        mc.updateBaseStatistics()  # means, variances, covariance, weight statistics; clears every per-parameter cache
    _hostlog("step: base statistics done")
    reset_caches(mc)
    my_params = parallel.partition_round_robin(list(range(mc.n)), world, rank)
    # N_eff is left to get2DDensities, which overlaps it with the 2D binning on a second stream; with several ranks the
    # ranges are exchanged first (the bin edges of every parameter are needed to start the binning) and each rank's
    # share of the N_eff values in a second, smaller all-gather once its kernels are through (parallel.NeffShare)
    mc.prepareParams(my_params, neff=False)
    if emulate:
        others = [j for j in range(mc.n) if j not in my_params]
        state = _REPLAY["rows"][others].copy()
        state[:, 1 + parallel.PARAM_STATE.index("N_eff_kde")] = np.nan
        parallel.unpack_param_state(mc, state)  # what the first all-gather would deliver

        def exchange(mc_):
            neff = _REPLAY["rows"][others][:, 1 + parallel.PARAM_STATE.index("N_eff_kde")]
            for j, v in zip(others, neff.tolist()):
                mc_.paramNames.names[j].N_eff_kde = v
    else:
        parallel.allgather_param_state(mc, my_params, mc.n, dist if world > 1 else None, torch_device, comm if world > 1 else None)

        def exchange(mc_):
            parallel.allgather_neff(mc_, my_params, mc_.n, dist, torch_device, comm)
    _hostlog("step: parameter state exchanged")
    t_part0 = time.perf_counter()
    _pair_index(pairs_all)
    if world == 1:
        my_pairs = _PAIR_INDEX["array"]  # nothing to deal out; the (npairs, 2) index array of the list, made once
    else:
        # whole tiles of the triangle per rank (a rank pre-bins only the columns its pairs touch), cost classes balanced;
        # GETDIST_AMD_PAIR_DEAL=class restores the round-robin deal over cost classes (every rank touches every column)
        if os.environ.get("GETDIST_AMD_PAIR_DEAL", "blocks") == "class":
            mine, _ = parallel.partition_pairs_by_class(pairs_all, pair_cost_classes(mc, pairs_all), world, rank)
        else:
            mine, _ = parallel.partition_pairs_by_column_blocks(_PAIR_INDEX["array"], pair_cost_classes(mc, pairs_all), world, rank, mc.n)
        my_pairs = _PAIR_INDEX["array"][mine]
        mc._neff_share = parallel.NeffShare(my_params, exchange)
        mc._neff_share.library_comm = comm is not None and not emulate
    if mc._timing:
        mc.timings["step.partition"] = mc.timings.get("step.partition", 0.0) + time.perf_counter() - t_part0
    _hostlog("step: pairs dealt")
    try:
        out = mc.get2DDensities(my_pairs)
    finally:
        share, mc._neff_share = getattr(mc, "_neff_share", None), None
        if share is not None:
            share.complete(mc)  # whatever this rank's share needed, it enters the step's collective exactly once
    _REPLAY["last_pairs"] = my_pairs  # the order of `out`
    if mc._timing:
        mc.timings["step.total"] = mc.timings.get("step.total", 0.0) + time.perf_counter() - t_step0
    return out


# ---- roofline of the batched 2D binning kernel ------------------------------------------------------------------------
def live_counter_traffic(mc, npairs, kernel):
    """
    HBM bytes per launch of the roofline kernel measured IN THIS RUN: two separate `rocprofv3 --kernel-trace --pmc` passes
    (FETCH_SIZE, then WRITE_SIZE: MI355X_MICROARCH.md -- one counter group per pass, never together with the tracing
    domains) over scripts/pmc_hist2d.py --only-u8, which launches the kernel on this very configuration.  bytes =
    FETCH_SIZE (KiB) x 2 (gfx950 tallies 128-byte requests at 64 B for 16-byte-per-lane streams) x 1024 + WRITE_SIZE x 1024.
    None when rocprofv3 is not on the box, a pass fails or times out: the caller falls back to the tracked counter file.
    """
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    want = kernel.split(" ")[0]
    vals = {}
    tmp = tempfile.mkdtemp(prefix="gdamd_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "h", "--", sys.executable,
                   os.path.join(ROOT, "scripts", "pmc_hist2d.py"), "--only-u8", "--nsamples=%d" % mc.numrows, "--nparams=%d" % mc.n]
            r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=150)
            if r.returncode != 0:
                return None
            got = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if row["Kernel_Name"].split("(")[0].replace("void ", "").startswith(want) and row["Counter_Name"] == counter:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None
            vals[counter] = sum(got) / len(got)
    except Exception:  # a profiler problem must not cost the bench line
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return traffic, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                     "scripts/pmc_hist2d.py --only-u8; FETCH_SIZE %.0f KiB x 2 + WRITE_SIZE %.0f KiB per %d-pair launch"
                     % (vals["FETCH_SIZE"], vals["WRITE_SIZE"], npairs))


def binning_kernel_roofline(mc, pairs_all, reps=5):
    """
    The O(N) kernel the path is built around -- the batched 2D binning of every base-grid pair (k_hist2d_u16) -- timed
    with HIP events on the library's stream.  `achieved` is its HBM traffic per launch (rocprofv3 PMC counters of this
    very launch shape, profiles/r02_pmc_hist2d.json: FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) over the
    measured time; when no counter file matches the configuration the unit-weight streaming model 16 N + 8 F^2 per
    density is used instead and says so.  The kernel is bound by LDS atomics (profiles/), not by HBM.
    """
    names = mc.paramNames.names
    F = mc.fine_bins_2D
    corr = mc.getCorrelationMatrix()
    sel = [p for p in pairs_all if abs(corr[p[1]][p[0]]) <= 0.866]
    use_u8 = mc.weights is None and F == 256 and hasattr(mc.ctx, "hist2d_prebinned8")
    if use_u8:
        wanted = {}
        for (a, b) in sel:
            for j in (a, b):
                fw, b0, _ = mc._bin_edges(names[j], F)
                wanted[j] = (b0, fw)
        use_u8 = mc._index_columns8(wanted)
    if use_u8:
        ix = [mc._idx_cols[(a, 256, "u8")][0] for a, b in sel]
        iy = [mc._idx_cols[(b, 256, "u8")][0] for a, b in sel]
        launch = lambda out: mc.ctx.hist2d_prebinned8(ix, iy, out=out)  # noqa: E731
        kernel = "k_hist2d_u8_pf (batched 2D binning of byte index columns, F=256, 16-bit packed LDS counters, v_perm addressing, 3-deep register ring of global loads)"
    else:
        ix, iy = [], []
        for (a, b) in sel:
            fwx, bx, _ = mc._bin_edges(names[a], F)
            fwy, by, _ = mc._bin_edges(names[b], F)
            ix.append(mc._index_column(a, F, bx, fwx))
            iy.append(mc._index_column(b, F, by, fwy))
        launch = lambda out: mc.ctx.hist2d_prebinned(ix, iy, F, out=out)  # noqa: E731
        kernel = "k_hist2d_u16 (batched 2D binning of pre-binned u16 index columns, 16-bit packed LDS counters)"
    out = mc.ctx.alloc(len(sel) * F * F * 8)
    launch(out)
    mc.ctx.sync()
    ms = []
    for _ in range(reps):
        mc.ctx.timer_start()
        launch(out)
        ms.append(mc.ctx.timer_stop_ms())
    out.free()
    t = max(float(np.median(ms)) * 1e-3, 1e-9)
    weighted = mc.weights is not None
    model_bytes = len(sel) * ((24.0 if weighted else 16.0) * mc.numrows + 8.0 * F * F)
    traffic = None
    source = "none"
    live = live_counter_traffic(mc, len(sel), kernel) if (use_u8 and not weighted and os.environ.get("GETDIST_AMD_LIVE_PMC", "1") == "1") else None
    if live is not None:
        traffic, source = live
    elif os.path.exists(PMC_FILE):
        pmc = json.load(open(PMC_FILE))
        if (pmc.get("N") == mc.numrows and pmc.get("n") == mc.n and pmc.get("F") == F and bool(pmc.get("weighted")) == weighted
                and pmc.get("pairs") and pmc.get("kernel", "").split(" ")[0].split("<")[0].replace("_pf", "") == kernel.split(" ")[0].replace("_pf", "")):
            traffic = float(pmc["hbm_bytes_per_launch"]) * len(sel) / pmc["pairs"]
            source = "TRACKED FILE (no live counter pass in this run) profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, scaled by pairs)" % os.path.basename(PMC_FILE)
    frac_model = model_bytes / t / 1e9 / HBM_PEAK_GBS
    if traffic is not None:
        achieved = traffic / t / 1e9
    else:
        achieved = model_bytes / t / 1e9
    # the roof the kernel actually sits under: random ds_add_u32 into a 128-KB table at this block shape, measured by
    # scripts/micro/lds_atomic_roof.hip (no memory traffic); one add per (sample, pair)
    lds = None
    if os.path.exists(LDS_ROOF_FILE):
        roof = json.load(open(LDS_ROOF_FILE))
        adds = float(len(sel)) * mc.numrows
        roof_ms = roof["ms_per_1p2e10_random_adds"] * adds / 1.2e10
        lds = dict(adds_per_launch=adds, roof_ms=roof_ms, frac_of_lds_atomic_roof=roof_ms / (t * 1e3),
                   source="profiles/r03_lds_atomic_roof.json (random ds_add_u32, 16 waves per CU, 128 KB table: %.2f lanes/clk/CU; "
                          "conflict-free: %.2f)" % (roof["lanes_per_clk_per_cu_random"], roof["conflict_free"]["lanes_per_clk_per_cu_at_2p4GHz"]))
    return dict(kernel=kernel, bound="lds-atomic", priced_against="hbm", lds_atomic_roof=lds, launches_pairs=len(sel), ms_per_launch=t * 1e3,
                achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                traffic_source=source, frac_counter=(None if traffic is None else traffic / t / 1e9 / HBM_PEAK_GBS),
                frac_model_unit_weight=frac_model, model_bytes=model_bytes,
                note="achieved = HBM counter bytes of this launch shape / HIP-event time (falls back to the streaming "
                     "model when no counter file matches). frac_model_unit_weight prices the same launch with the "
                     "streaming model of SURVEY 8d for unit weights (x, y read once as fp64 + the F x F grid = 16N + 8F^2 "
                     "per density); it exceeds 1 because the kernel reads 2-byte pre-binned indices shared by 49 pairs "
                     "each instead of the fp64 columns.")


# ---- CPU baseline ----------------------------------------------------------------------------------------
def _cpu_noop(k):
    return k


def _peak_rss_mb():
    """Peak resident set of THIS process image (VmHWM; ru_maxrss would carry the parent's size across fork + exec)."""
    try:
        for line in open("/proc/self/status"):
            if line.startswith("VmHWM:"):
                return int(line.split()[1]) / 1024.0
    except OSError:
        pass
    return 0.0


def _verdict_record(v):
    """What is reported per loose pair: the criterion ("inside" = between the oracle ensemble's extremes, "spread" = within
    0.25 of its spread beyond them, "amise" = as good in the reference's own objective) and the perturbation scale that admitted
    the device's bandwidth triple under the frozen rules (gate: slack 0.25 since round 6), the distance to the nearest member
    of the oracle's ensemble, and whether the wider slack of rounds 3-5 (1.0) would have admitted it (reported only)."""
    return dict(admitted=bool(v["ok"]), admitted_by=v.get("admitted_by"), perturbation_scale=v["scale"],
                nearest_member_rel=v.get("nearest_member"), excess_over_spread=v["excess"], amise_excess=v["amise_excess"],
                amise_range=v["amise_range"], between_the_ensemble_extremes=bool(v.get("inside")),
                admitted_at_slack_1p0_reported_only=bool(v.get("ok_at_slack_1")), members=v["members"])


def _cpu_task(task):
    """Worker: one oracle task on memory-mapped sample columns.  kinds: 'prep' (ranges + N_eff of one parameter),
    'pair' (one 2D density with its parameters' N_eff prepared beforehand, as in a triangle) and 'triangle' (a share of
    the full triangle at small N, parameter state cached in the worker)."""
    import warnings

    from oracle import kde_oracle as ko

    warnings.simplefilter("ignore")
    import logging

    logging.disable(logging.WARNING)
    kind = task["kind"]
    s = np.load(task["path"], mmap_mode="r")
    names, ranges = task["names"], task["ranges"]
    if kind == "prep":
        j = task["j"]
        orc = ko.OracleSamples(np.array(s[:, [j]]), names=[names[j]], ranges={k: v for k, v in ranges.items() if k == names[j]})
        t0 = time.perf_counter()
        orc.init_param(0)
        neff = orc.neff_1d(0)
        # (the parameter's prepared state -- ranges, limit flags, sigma_range, N_eff: plain scalars -- travels to the census tasks)
        state = {k: v for k, v in vars(orc.pars[0]).items() if k != "name"}
        return dict(kind=kind, j=j, seconds=time.perf_counter() - t0, neff=float(neff), state=state)
    if kind == "pair":
        a, b = task["pair"]
        sub = [names[a], names[b]]
        orc = ko.OracleSamples(np.array(s[:, [a, b]]), names=sub, ranges={k: v for k, v in ranges.items() if k in sub})
        for k in (0, 1):
            orc.init_param(k)
            orc.neff_1d(k)
        tr = {}
        t0 = time.perf_counter()
        o = orc.density_2d(0, 1, trace=tr)
        dt = time.perf_counter() - t0
        out = dict(kind=kind, pair=(a, b), seconds=dt, branch=tr.get("branch"), bw=(tr.get("hx"), tr.get("hy"), tr.get("c")),
                   F=int(o["P"].shape[0]))
        if task.get("want_grid"):
            out["P"] = o["P"]
            if "p_13" in tr:  # a TNC pair: the oracle's own spread for rounding-equal inputs, and where the device's triple lies
                psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
                if task.get("gpu_kopt") is not None:
                    v = ko.judge_triple(np.asarray(task["gpu_kopt"])[8:11], psi, tr["opt_N"], tr["opt_corr"])
                    out["chaotic"] = (v["moved"] > 1e-6, v["moved"])
                    out["inside_oracle_spread"], out["excess"], out["amise_ok"] = v["inside"], v["excess"], v["amise_ok"]
                    out["within_spread_slack_0p25"] = v["within_slack"]
                    out["ensemble_perturbation"] = v["scale"]
                    out["verdict"] = _verdict_record(v)
                else:
                    ens = ko.get_h_ensemble(psi, tr["opt_N"], tr["opt_corr"])
                    moved = float(np.max(np.abs(ens - ens[0])) / np.max(np.abs(ens[0])))
                    out["chaotic"] = (moved > 1e-6, moved)
        return out
    if kind == "census_pair":
        # one pair of the full-size census: its two columns only (160 MB at N = 1e7), the parameters' N_eff preset from the
        # preparation stage (as inside a triangle), the oracle's grid against the GPU grid read from the flat file
        a, b = task["pair"]
        sub = [names[a], names[b]]
        orc = ko.OracleSamples(np.array(s[:, [a, b]]), names=sub, ranges={k: v for k, v in ranges.items() if k in sub})
        for k, j in enumerate((a, b)):
            if task.get("state") is not None:  # prepared once per parameter by the CPU baseline's preparation stage
                vars(orc.pars[k]).update(task["state"][j])
            else:
                orc.init_param(k)
            orc.pars[k].N_eff_kde = task["neff"][j]
        tr = {}
        t0 = time.perf_counter()
        P = orc.density_2d(0, 1, trace=tr)["P"]
        seconds = time.perf_counter() - t0
        F = task["gpu_F"]
        row = dict(pair=(a, b), F=int(P.shape[0]), branch=tr.get("branch"), shape_ok=bool(P.shape[0] == F), tnc="p_13" in tr,
                   seconds=seconds, max_rss_mb=_peak_rss_mb())
        if row["shape_ok"]:
            G = np.asarray(np.load(task["gpu_path"], mmap_mode="r")[task["gpu_off"]:task["gpu_off"] + F * F]).reshape(F, F)
            row["err"] = float(np.max(np.abs(G - P)))
            kopt = task["gpu_kopt"]
            if row["err"] > 1e-6 and row["tnc"] and kopt is not None:
                psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
                v = ko.judge_triple(np.asarray(kopt)[8:11], psi, tr["opt_N"], tr["opt_corr"])
                row["oracle_moves_by"] = v["moved"]
                row["inside_oracle_spread"], row["excess"], row["amise_ok"] = v["inside"], v["excess"], v["amise_ok"]
                row["within_spread_slack_0p25"] = v["within_slack"]
                row["ensemble_perturbation"] = v["scale"]
                row["verdict"] = _verdict_record(v)
        return row
    # share of a full triangle: this worker's pairs, parameter state cached across them.  The CPU time is the oracle's
    # alone; afterwards (outside the timed span) every grid is compared with the GPU's grid of the same pair, read from
    # the flat file the GPU run left in shared memory, and a pair above 1e-6 is put to the oracle-ensemble test.
    # task["cols"]: only these columns are copied out of the mapped file (the full-size census: a worker's share touches
    # a tile of the triangle, about ten columns); pair indices are remapped to them
    pairs_here = task["pairs"]
    if task.get("cols") is not None:
        cols = list(task["cols"])
        at_col = {c: k for k, c in enumerate(cols)}
        sub_names = [names[c] for c in cols]
        orc = ko.OracleSamples(np.asfortranarray(s[:, cols]), names=sub_names, ranges={k: v for k, v in ranges.items() if k in sub_names})
        pairs_here = [(at_col[a], at_col[b]) for a, b in task["pairs"]]
    else:
        orc = ko.OracleSamples(np.array(s), names=names, ranges=ranges)
    t0 = time.perf_counter()
    grids, traces = [], []
    for a, b in pairs_here:
        for k in (a, b):
            if orc.pars[k].N_eff_kde is None:
                orc.init_param(k)
                orc.neff_1d(k)
        tr = {}
        grids.append(orc.density_2d(a, b, trace=tr)["P"])
        traces.append(tr)
    seconds = time.perf_counter() - t0
    gpu = np.load(task["gpu_path"], mmap_mode="r")
    rows = []
    for (a, b), P, tr, off, F, kopt in zip(task["pairs"], grids, traces, task["gpu_offsets"], task["gpu_F"], task["gpu_kopt"]):
        row = dict(pair=(a, b), F=int(P.shape[0]), branch=tr.get("branch"), shape_ok=bool(P.shape[0] == F), tnc="p_13" in tr)
        if row["shape_ok"]:
            G = np.asarray(gpu[off:off + F * F]).reshape(F, F)
            row["err"] = float(np.max(np.abs(G - P)))
            row["sum_rel"] = float(abs(np.sum(G) - np.sum(P)) / np.sum(P))
            if row["err"] > 1e-6 and row["tnc"] and kopt is not None:
                psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
                v = ko.judge_triple(kopt[8:11], psi, tr["opt_N"], tr["opt_corr"])
                row["oracle_moves_by"] = v["moved"]
                row["inside_oracle_spread"], row["excess"], row["amise_ok"] = v["inside"], v["excess"], v["amise_ok"]
                row["within_spread_slack_0p25"] = v["within_slack"]
                row["ensemble_perturbation"] = v["scale"]
                row["verdict"] = _verdict_record(v)
        rows.append(row)
    return dict(kind=kind, seconds=seconds, rows=rows)


def _census_summary(rows, has_limits, names):
    """Per-class table and the loose pairs (with their verdict records) of an all-pairs comparison."""
    census = {}
    for row in rows:
        a, b = row["pair"]
        key = "%s/%d/%d" % (row["branch"], int(bool(has_limits[a])) + int(bool(has_limits[b])), row["F"])
        c = census.setdefault(key, dict(pairs=0, errs=[], loose=0))
        c["pairs"] += 1
        c["errs"].append(row.get("err", float("inf")))
        c["loose"] += row.get("err", float("inf")) > 1e-6
    loose_rows = [r for r in rows if r.get("err", float("inf")) > 1e-6]
    return dict(
        gate="max|dP| of the max-normalised grids, GPU vs oracle, every pair of the triangle",
        pairs_compared=len(rows), grid_shapes_equal=int(sum(r["shape_ok"] for r in rows)),
        pairs_within_1e_6=int(sum(r.get("err", 9.0) <= 1e-6 for r in rows)),
        pairs_above_1e_6=len(loose_rows), worst_abs_dP=float(max([r.get("err", 0.0) for r in rows] or [0.0])),
        loose_pairs_that_use_tnc=int(sum(bool(r["tnc"]) for r in loose_rows)),
        loose_pairs_chaotic_in_the_oracle=int(sum(r.get("oracle_moves_by", 0.0) > 1e-6 for r in loose_rows)),
        # the admission rule (oracle.kde_oracle.judge_triple; gate = slack 0.25 of the ensemble's spread since round 6):
        loose_pairs_between_the_oracle_ensembles_extremes=int(sum(bool(r.get("inside_oracle_spread")) for r in loose_rows)),
        loose_pairs_within_0p25_of_the_spread_or_as_good_in_amise=int(sum(bool(r.get("within_spread_slack_0p25") or r.get("amise_ok")) for r in loose_rows)),
        loose_pairs_admitted=int(sum(bool((r.get("verdict") or {}).get("admitted")) for r in loose_rows)),
        worst_excess_over_oracle_spread=float(max([r.get("excess", 0.0) for r in loose_rows] or [0.0])),
        loose_pairs=[dict(pair=[names[r["pair"][0]], names[r["pair"][1]]], max_abs_dP=r.get("err"), verdict=r.get("verdict"))
                     for r in loose_rows],
        per_class={k: dict(pairs=v["pairs"], above_1e_6=int(v["loose"]), max_abs_dP=float(np.max(v["errs"])),
                           median_abs_dP=float(np.median(v["errs"]))) for k, v in sorted(census.items())})


def _host_memory_budget():
    """Bytes this process tree may safely use: the smaller of the cgroup limit (v2 memory.max / v1 limit_in_bytes -- a
    container's ceiling, which /proc/meminfo does not show) and MemAvailable.  None when neither can be read."""
    vals = []
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            txt = open(path).read().strip()
            if txt != "max" and int(txt) < (1 << 60):
                used = 0
                for upath in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                    try:
                        used = int(open(upath).read().strip())
                        break
                    except (OSError, ValueError):
                        pass
                vals.append(int(txt) - used)
        except (OSError, ValueError):
            pass
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                vals.append(int(line.split()[1]) * 1024)
    except OSError:
        pass
    return min(vals) if vals else None


def full_size_census(args, s_path, names, ranges, pairs_all, dens, has_limits, tmp, cores, neff_by_param, state_by_param=None):
    """EVERY pair of the triangle at full N against the oracle (review item, round 4: the driver-run parity was a sample of
    31), when the host has the cores for it: one task per pair -- its two columns only, the parameters' N_eff taken from
    the preparation stage of the CPU baseline -- the GPU grids in one flat file in shared memory.  A worker's peak is
    ~1 GB at N = 1e7; the pool is sized so that all workers together stay below a THIRD of what the host / the container
    allows (cgroup limit and MemAvailable), and the stage has a wall-clock budget.  Returns None when it does not run, a
    dict saying why when it gives up."""
    import multiprocessing as mp

    min_cores = int(os.environ.get("GETDIST_AMD_CENSUS_MIN_CORES", "64"))  # (lowered by the CPU smoke of this function only)
    if cores < min_cores or os.environ.get("GETDIST_AMD_FULL_CENSUS", "1") != "1":
        return None
    N = int(np.load(s_path, mmap_mode="r").shape[0])
    budget = _host_memory_budget()
    per_worker = 12 * N * 8 + 400e6  # two columns, the oracle's copies and sort temporaries, the interpreter (measured: max_rss)
    # (the oracle's sorts are memory-bound: 56 workers took 5.2 s per pair where 32 take 2.7 -- half the cores, at most 48)
    workers = int(os.environ.get("GETDIST_AMD_CENSUS_WORKERS", "0")) or int(min(cores // 2, 48, len(pairs_all)))
    if budget is None:
        workers = min(workers, 8)
    else:
        workers = int(min(workers, (0.35 * budget) // per_worker))  # (measured peak of a worker: 1.0 GB at N = 1e7, VmHWM)
    if workers < 2:
        return dict(ran=False, reason="not enough free host memory for the census pool (budget %s bytes)" % budget)
    sizes = np.array([d.P.size for d in dens], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    flat = np.empty(int(offs[-1]))
    for d, o in zip(dens, offs[:-1]):
        flat[o:o + d.P.size] = d.P.ravel()
    path_gpu = os.path.join(tmp, "gpu_grids_full.npy")
    np.save(path_gpu, flat)
    del flat
    base = dict(kind="census_pair", path=s_path, names=list(names), ranges=dict(ranges), gpu_path=path_gpu, neff=dict(neff_by_param),
                state=state_by_param)
    tasks = [dict(base, pair=pr, gpu_off=int(offs[k]), gpu_F=int(dens[k].P.shape[0]),
                  gpu_kopt=None if dens[k].kopt is None else np.asarray(dens[k].kopt)) for k, pr in enumerate(pairs_all)]
    # longest first: the up-scaled grids (the oracle's frames grow with F^2), then the sheared branch, then the rest
    branch_of = {pr: d.bandwidth_branch for pr, d in zip(pairs_all, dens)}
    tasks.sort(key=lambda t: (-t["gpu_F"], branch_of[t["pair"]] != "A"))
    limit = int(os.environ.get("GETDIST_AMD_CENSUS_MAX_PAIRS", "0"))  # (probing a new host: the first K pairs only)
    if limit:
        tasks = tasks[:limit]
    saved_env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved_env:
        os.environ[k] = "1"
    t0 = time.perf_counter()
    try:
        # (the single-thread settings stay in the environment until the pool is closed: a worker spawned later -- after a
        # crash, or with maxtasksperchild -- would otherwise start with one BLAS / OpenMP thread per core)
        with mp.get_context("spawn").Pool(workers) as pool:
            try:
                rows = pool.map_async(_cpu_task, tasks, chunksize=1).get(timeout=args.census_budget_s)
            except mp.TimeoutError:
                return dict(ran=True, abandoned=True, reason="wall-clock budget of %.0f s exceeded" % args.census_budget_s,
                            workers=workers)
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = _census_summary(rows, has_limits, names)
    out.update(ran=True, N=N, workers=workers, wall_s=round(time.perf_counter() - t0, 1),
               cpu_core_seconds=round(sum(r["seconds"] for r in rows), 1),
               worker_max_rss_mb=round(max(r.get("max_rss_mb", 0.0) for r in rows), 0),
               host_memory_budget_gb=None if budget is None else round(budget / 1e9, 1),
               note="every pair of the timed triangle at full size against the oracle; one task per pair (two columns), N_eff "
                    "from the preparation stage; the pool is sized to a quarter of the host / container memory")
    return out


def cpu_baseline_and_parity(args, mc, s, names, ranges, pairs_all, dens):
    """the oracle on this host's cores next to the GPU numbers, and full-size parity of the sample."""
    import multiprocessing as mp
    import tempfile

    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    cores = os.cpu_count() or 1
    workers = args.cpu_workers or min(cores, 32)
    N, n = s.shape
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="gdamd_bench_", dir=shm)
    path_full = os.path.join(tmp, "full.npy")
    np.save(path_full, np.asfortranarray(s))
    # class census from the GPU run: (bandwidth branch, #bounded parameters, grid size)
    par = mc.paramNames.names
    klass = {}
    for (a, b), d in zip(pairs_all, dens):
        key = "%s/%d/%d" % (d.bandwidth_branch, int(bool(par[a].has_limits)) + int(bool(par[b].has_limits)), d.P.shape[0])
        klass.setdefault(key, []).append((a, b))
    # up to 3 pairs per class, 6 from the classes that hold a hundred pairs or more (spread over the class, not its head)
    # (a host without the cores for the all-pairs census below takes every sheared -- branch A -- pair instead: they are
    # the ones whose bandwidth goes through the re-binned grid)
    census_will_run = cores >= int(os.environ.get("GETDIST_AMD_CENSUS_MIN_CORES", "64")) and os.environ.get("GETDIST_AMD_FULL_CENSUS", "1") == "1"
    sample = []
    for key, members in sorted(klass.items()):
        want = 6 if len(members) >= 100 else 3
        if key.startswith("A/") and not census_will_run:
            want = len(members)
        step = max(1, len(members) // want)
        sample += [(key, pr) for pr in members[::step][:want]]
    base = dict(path=path_full, names=list(names), ranges=dict(ranges))
    kopt_of = {pr: (None if d.kopt is None else np.asarray(d.kopt)) for pr, d in zip(pairs_all, dens)}
    tasks = [dict(base, kind="prep", j=j) for j in range(n)] + \
            [dict(base, kind="pair", pair=pr, want_grid=True, gpu_kopt=kopt_of[pr]) for _, pr in sample]
    ctx = mp.get_context("spawn")
    # the workers are single-threaded: the BLAS / OpenMP pools read these variables when numpy is first imported in
    # the child, so they must be in the environment the children are spawned with
    saved_env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved_env:
        os.environ[k] = "1"
    t0 = time.perf_counter()
    pool_cm = ctx.Pool(workers)
    for k, v in saved_env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    with pool_cm as pool:
        pool.map(_cpu_noop, range(workers * 2), chunksize=1)  # start-up (interpreter + numpy import) is not CPU-baseline time
        t0 = time.perf_counter()
        # longest tasks first keeps the pool busy to the end
        res = pool.map_async(_cpu_task, sorted(tasks, key=lambda t: 0 if t["kind"] == "pair" else 1),
                             chunksize=1).get(timeout=args.cpu_budget_s)
        wall_sample = time.perf_counter() - t0
        prep = {r["j"]: r["seconds"] for r in res if r["kind"] == "prep"}
        neff_by_param = {r["j"]: r["neff"] for r in res if r["kind"] == "prep"}
        state_by_param = {r["j"]: r["state"] for r in res if r["kind"] == "prep"}
        pair_res = {tuple(r["pair"]): r for r in res if r["kind"] == "pair"}
        by_class = {}
        for key, pr in sample:
            by_class.setdefault(key, []).append(pair_res[pr]["seconds"])
        t_preps = sum(prep.values())
        t_pairs = sum(len(klass[key]) * float(np.mean(v)) for key, v in by_class.items())
        cpu_seconds_triangle = t_preps + t_pairs  # one core doing everything
        task_seconds = t_preps + sum(sum(v) for v in by_class.values())
        pool_efficiency = task_seconds / (wall_sample * workers)
        value_pool = len(pairs_all) / (cpu_seconds_triangle / (workers * min(1.0, pool_efficiency)))
        # ---- parity of the stratified sample at full size
        dens_by_pair = dict(zip(pairs_all, dens))
        parity = {}
        loose = []
        for key, pr in sample:
            r, d = pair_res[pr], dens_by_pair[pr]
            err = float(np.max(np.abs(d.P - r["P"]))) if d.P.shape == r["P"].shape else float("inf")
            bw_err = float(np.max(np.abs(np.array(d.bandwidth) - np.array(r["bw"], dtype=float)))
                           / max(abs(r["bw"][0]), abs(r["bw"][1])))
            ent = parity.setdefault(key, dict(pairs_in_class=len(klass[key]), checked=0, max_abs_dP=0.0, max_bandwidth_rel_err=0.0))
            ent["checked"] += 1
            ent["max_abs_dP"] = max(ent["max_abs_dP"], err)
            ent["max_bandwidth_rel_err"] = max(ent["max_bandwidth_rel_err"], bw_err)
            if err > 1e-6:
                chaotic = r.get("chaotic", (False, 0.0))
                loose.append(dict(pair=[names[pr[0]], names[pr[1]]], klass=key, max_abs_dP=err, bandwidth_rel_err=bw_err,
                                  oracle_chaotic=bool(chaotic[0]), oracle_moves_by=float(chaotic[1]),
                                  inside_oracle_spread=bool(r.get("inside_oracle_spread", False)),
                                  as_good_in_amise=bool(r.get("amise_ok", False)),
                                  excess_over_oracle_spread=float(r.get("excess", 0.0)), verdict=r.get("verdict")))
        parity_block = dict(N=int(N), tolerance=1e-6, classes=parity, n_pairs_checked=len(sample), n_pairs_on_loose_gate=len(loose),
                            loose_pairs=loose,
                            n_loose_pairs_chaotic_in_the_oracle=int(sum(e_["oracle_chaotic"] for e_ in loose)),
                            n_loose_pairs_inside_the_oracle_spread=int(sum(e_["inside_oracle_spread"] for e_ in loose)),
                            note="max|dP| of the GPU grid against the oracle grid (both max-normalised); a pair above the "
                                 "tolerance is examined: the oracle's get_h on 24 copies of its own functionals perturbed by "
                                 "+-1..12e-15 (its spread), and whether the device's raw (hx, hy, c) lies inside that spread")
        # ---- single-process sample (the reference as shipped is one process): two preparations + two pairs
        t0 = time.perf_counter()
        single = [_cpu_task(dict(base, kind="prep", j=min(5, n - 1))), _cpu_task(dict(base, kind="pair", pair=sample[0][1]))]
        wall_single = time.perf_counter() - t0
        sp_prep = float(np.mean([r["seconds"] for r in single if r["kind"] == "prep"]))
        sp_ratio = float(np.mean([r["seconds"] / pair_res[tuple(r["pair"])]["seconds"] for r in single if r["kind"] == "pair"]))
        value_single = len(pairs_all) / (n * sp_prep + t_pairs * sp_ratio)
        # ---- un-extrapolated cross-check: the full triangle at small N on the same cores and on the GPU
        small = None
        if args.cpu_small_n and args.cpu_small_n < N:
            s2, w2, names2, ranges2 = synth.config_c3(args.cpu_small_n, n)
            path_small = os.path.join(tmp, "small.npy")
            np.save(path_small, np.asfortranarray(s2))
            mc2 = MCSamples(samples=s2, weights=w2, names=names2, ranges=ranges2, device=mc._device,
                            _context_factory=mc._context_factory)  # the HIP context class (a stand-in only in the CPU suite)
            for _ in range(3):  # plans, page-locked result blocks and the second set of device blocks exist afterwards
                mc2.get2DDensities(pairs_all)
                reset_caches(mc2)
            mc2.ctx.sync()
            mc2.ctx.copy_sync()
            t0 = time.perf_counter()
            mc2.updateBaseStatistics()
            d2 = mc2.get2DDensities(pairs_all)
            d2[-1].P  # waits for the result copies of this call
            mc2.ctx.sync()
            wall_gpu = time.perf_counter() - t0
            # the GPU grids of every pair in one flat file the CPU workers map
            sizes = np.array([d.P.size for d in d2], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum(sizes)])
            flat = np.empty(int(offs[-1]))
            for d, o in zip(d2, offs[:-1]):
                flat[o:o + d.P.size] = d.P.ravel()
            path_gpu = os.path.join(tmp, "gpu_grids.npy")
            np.save(path_gpu, flat)
            del flat
            at = {pr: k for k, pr in enumerate(pairs_all)}
            shares = [pairs_all[k::workers] for k in range(workers)]
            t0 = time.perf_counter()
            tri = pool.map_async(_cpu_task, [dict(kind="triangle", path=path_small, names=list(names2), ranges=dict(ranges2),
                                                  pairs=sh, gpu_path=path_gpu, gpu_offsets=[int(offs[at[pr]]) for pr in sh],
                                                  gpu_F=[int(d2[at[pr]].P.shape[0]) for pr in sh],
                                                  gpu_kopt=[None if d2[at[pr]].kopt is None else np.asarray(d2[at[pr]].kopt)
                                                            for pr in sh])
                                             for sh in shares if sh], chunksize=1).get(timeout=args.cpu_budget_s)
            wall_cpu = max(r["seconds"] for r in tri)  # the slowest worker's oracle time = the pool's wall time
            rows = [row for r in tri for row in r["rows"]]
            par2 = mc2.paramNames.names
            small = dict(N=int(args.cpu_small_n), pairs=len(pairs_all), cpu_wall_s=round(wall_cpu, 2), cpu_workers=workers,
                         cpu_densities_per_s=round(len(pairs_all) / wall_cpu, 2), cpu_core_seconds=round(sum(r["seconds"] for r in tri), 1),
                         gpu_wall_s=round(wall_gpu, 4), gpu_densities_per_s=round(len(pairs_all) / wall_gpu, 1),
                         gpu_over_cpu_pool=round(wall_cpu / wall_gpu, 1),
                         parity_census=_census_summary(rows, [bool(p.has_limits) for p in par2], list(names2)))
            mc2.ctx.close()
    # ---- every pair of the timed triangle at full size (hosts with >= 64 cores; behind the clock)
    try:
        full = full_size_census(args, path_full, names, ranges, pairs_all, dens, [bool(p.has_limits) for p in par], tmp, cores,
                                neff_by_param, state_by_param)
    except Exception as exc:  # the sample above stands on its own
        full = dict(ran=False, reason="census failed: %r" % (exc,))
    parity_block["full_size_census"] = full
    try:
        import shutil

        shutil.rmtree(tmp)
    except OSError:
        pass
    cpu = dict(value=value_pool, unit="densities/s", cores=workers, kind="port",
               sample="oracle (numpy/scipy restatement of the reference, validated against it) at N=%d: all %d per-parameter "
                      "preparations (%.1f core-s) + %d pairs stratified over the %d (branch/#bounded/F) classes of the census "
                      "(%.1f core-s), in a pool of %d single-threaded workers (%.1f s wall, %.0f %% busy); triangle extrapolated "
                      "with the per-class pair counts as core-seconds / (workers x busy fraction); host has %d logical cores"
                      % (N, n, t_preps, len(sample), len(by_class), task_seconds - t_preps, workers, wall_sample,
                         100 * pool_efficiency, cores),
               single_process_value=value_single,
               single_process_sample="one preparation + one pair (with its two preparations) run alone in this process with the "
                                     "default BLAS threads (%.1f s); the reference as shipped is single-process" % wall_single,
               cpu_core_seconds_per_triangle=round(cpu_seconds_triangle, 1),
               per_class_pair_seconds={k: round(float(np.mean(v)), 3) for k, v in by_class.items()},
               class_census={k: len(v) for k, v in sorted(klass.items())},
               full_triangle_small_n=small)
    return cpu, parity_block


def main():
    args = parse()
    spawn_ranks_if_needed(args)
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
        world = dist.get_world_size()  # from the communicator, not from an environment default
    if world != max(args.gpus, 1) and not args.emulate_world:
        raise SystemExit("bench.py: --gpus %d but the communicator has %d rank(s): launch with "
                         "`python bench.py --gpus N` or torch.distributed.run --nproc-per-node N" % (args.gpus, world))

    import logging

    logging.getLogger().setLevel(logging.ERROR)  # the reference's "fine_bins_2D not large enough" warnings, 170 per step
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    t0 = time.perf_counter()
    s, w, names, ranges = synth.config_c3(args.nsamples, args.nparams)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    extra = {}
    if args.context_factory:  # CPU suite only (tests/fake_ctx.py); see parse()
        import importlib

        mod, attr = args.context_factory.split(":")
        extra["_context_factory"] = getattr(importlib.import_module(mod), attr)
    share = None
    if dist is not None and args.backend == "nccl" and os.environ.get("GETDIST_AMD_COMM", "lib") == "lib":
        # the step's collectives through the C ABI: RCCL on the library's own stream (torch.distributed only carries the
        # 128-byte id at start-up, and the barriers / the max over ranks around the timed region).  The communicator is made
        # by the FIRST upload (all-or-nothing over the ranks, every stage under a watchdog: parallel.init_library_comm), which
        # then sends only this rank's block of columns over PCIe and receives the others over xGMI (parallel.ColumnShare)
        from getdist_amd import parallel

        share = parallel.ColumnShare(dist, rank, world, torch_device)
        extra["column_share"] = share
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, device=local_rank, **extra)
    t_ctor = time.perf_counter() - t0
    pairs_all = synth.triangle_pairs(args.nparams)
    comm = share.comm if share is not None else None  # None on every rank if any rank cannot use the library's collectives

    def barrier():
        mc.ctx.sync()
        mc.ctx.copy_sync()  # the result copies of the last step (delivered while the next step computes) have landed
        if dist is not None:
            import torch

            dist.barrier()
            if torch_device is not None:
                torch.cuda.synchronize()

    if args.emulate_world:
        prepare_replay(mc, args.emulate_world)
    dens = None
    for _ in range(args.warmup):
        dens = one_step(mc, pairs_all, dist, rank, world, torch_device, args.emulate_world, comm)  # held like the timed results
        if dens:
            dens[-1].P  # ... and delivered like them (the library schedules a call behind undelivered results for throughput)
    if args.warmup > 0:
        mc.ctx.reserve_pinned_twin()  # result buffers for "previous step still referenced" + "current step"
        if getattr(mc, "_twin", None) is not None:
            mc._twin.ctx.reserve_pinned_twin()  # a rank's share is convolved on both contexts' streams
    # The cyclic collector's passes walk the live objects (a step's 1225 result objects, the interpreter's modules):
    # 5-10 ms every eighth step on this host, all of it in front of a kernel launch, and they find nothing -- the path
    # creates no reference cycles (every per-step object is freed by its reference count; checked with
    # gc.DEBUG_SAVEALL).  A caller that loops over batched calls does what this does: collect once, freeze what is
    # alive, and keep the collector off while it loops.
    import gc

    gc.collect()
    gc.freeze()
    gc.disable()
    barrier()
    mc.timings = {}
    prof = None
    if args.profile:
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    # ---- the timed region: K triangles, each one DELIVERED before the next starts (t spans "columns resident in
    # HBM" -> "all 1225 normalised grids available on host"; the first read of a grid waits for the step's result copies and
    # checks every grid's status).  This is the headline.
    t0 = time.perf_counter()
    step_returned = []
    for _ in range(args.steps):
        dens = one_step(mc, pairs_all, dist, rank, world, torch_device, args.emulate_world, comm)
        if dens:
            dens[-1].P
        step_returned.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    # ---- the same K steps as a stream of triangles: the PCIe copy of step k's grids (642 MB) lands while step k + 1 computes
    # (what a caller that keeps asking for triangles gets: reported beside the headline, never as `value`)
    pipelined_ms = None
    if prof is None:
        ts = time.perf_counter()
        for _ in range(args.steps):
            dens = one_step(mc, pairs_all, dist, rank, world, torch_device, args.emulate_world, comm)
        if dens:
            dens[-1].P
        barrier()
        pipelined_ms = (time.perf_counter() - ts) / args.steps * 1e3
    serial_ms = elapsed / args.steps * 1e3
    gc.enable()
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(30)
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device=torch_device or "cpu")
        every = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    npairs = len(pairs_all)
    value = npairs * args.steps / elapsed  # with --emulate-world: what W ranks would deliver if all took this long

    if rank == 0:
        line = {
            "metric": "2D KDE densities/sec (triangle, %d params, %s samples)" % (args.nparams, "{:.0e}".format(args.nsamples)),
            "value": value, "unit": "densities/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_single_triangle_latency": serial_ms,
            # the stream-of-triangles rate (result copies of step k under step k + 1): a second figure, not the headline
            "ms_per_step_pipelined": pipelined_ms,
            "value_pipelined": (None if pipelined_ms is None else npairs / (pipelined_ms * 1e-3)),
            # host time from one delivered triangle to the next
            "ms_between_delivered_triangles": [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_returned[:-1], step_returned)],
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic (seeded block recipe, SURVEY.md 8d C3)",
            "config": {"workload": "C3: 2D KDE triangle, %d params (%d pairs), N=%d unit-weight samples, base fine_bins_2D=256, "
                                   "default settings; every step includes base statistics (means, variances, covariance), "
                                   "per-parameter prep, bandwidth selection (fixed point + TNC on the device) and D2H of all grids"
                                   % (args.nparams, npairs, args.nsamples),
                       "parallelism": "pairs partitioned over %d GPU(s), samples replicated" % world,
                       "timed_region": "K triangles one after the other, each with all its grids on the host before the next "
                                       "starts (SURVEY 8d's t); value = pairs / that time.  value_pipelined / "
                                       "ms_per_step_pipelined = the same K steps with the PCIe copy of a step's grids (642 MB) "
                                       "landing while the next step computes",
                       "setup_s": {"generate": round(t_gen, 2), "construct_upload_basestats": round(t_ctor, 2)}},
        }
        if dist is not None:
            line["ranks"] = world
            line["backend"] = "%s%s" % (args.backend, " (RCCL)" if args.backend == "nccl" else "")
            line["collectives"] = ("libgdhip gd_comm_* (ncclAllGather / ncclAllReduce on the library's stream)" if comm is not None
                                   else "torch.distributed")
            line["ms_per_step_by_rank"] = [round(v, 3) for v in per_rank_ms]
            line["sample_distribution"] = ("each rank uploads n/W columns (%.2f GB on rank 0), the blocks are broadcast over xGMI "
                                           "(gd_comm_share_columns)" % (share.bytes_uploaded / 1e9) if comm is not None else
                                           "every rank uploads the full set")
        if args.emulate_world:
            line["emulated_world"] = args.emulate_world
            line["n_gpus"] = args.emulate_world
        if world == 1 and not args.emulate_world:
            line["roofline"] = binning_kernel_roofline(mc, pairs_all)
            if not args.no_cpu_baseline:
                order = {tuple(pr): k for k, pr in enumerate(np.asarray(_REPLAY["last_pairs"]).tolist())}
                try:
                    cpu, parity = cpu_baseline_and_parity(args, mc, s, names, ranges, pairs_all,
                                                          [dens[order[pr]] for pr in pairs_all])
                    line["cpu_baseline"] = cpu
                    line["parity"] = parity
                except Exception as exc:  # never lose the measured GPU line to a host-side problem
                    line["cpu_baseline"] = dict(value=None, unit="densities/s", cores=0, kind="port",
                                                sample="CPU baseline failed: %r" % (exc,))
        if mc._timing:
            line["phase_seconds_total"] = {k: round(v, 4) for k, v in sorted(mc.timings.items())}
        assert len(dens) > 0 and all(d is not None for d in dens)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
