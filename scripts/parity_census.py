"""
Full-size parity census: EVERY pair of the C3 triangle (50 parameters, 1225 pairs, N = 1e7 by default) computed on the
GPU through the product path and by the oracle (numpy / scipy restatement of the reference, oracle/kde_oracle.py) in a
pool of host workers, compared pixel by pixel.  For every pair above 1e-6 the oracle's own sensitivity is examined: its
get_h on 24 copies of its OWN functionals perturbed by +-1..12e-15 (the ensemble; widened to 1e-14 .. 1e-12 only if the
triple is not admitted, ko.judge_triple), whether the device's bandwidth
triple lies inside that spread, how far it is from the NEAREST ensemble member, and whether its AMISE is as good.

    python scripts/parity_census.py [--nsamples 10000000] [--nparams 50] [--workers 0] [--weighted] [--out profiles/...json]

Test / evidence infrastructure (it imports the oracle); not part of the product or of bench.py's timed region.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _quiet():
    import logging
    import warnings

    warnings.simplefilter("ignore")
    logging.disable(logging.WARNING)


def prep_task(t):
    """N_eff of one parameter (mcsamples.py:1230-1235) by the oracle."""
    _quiet()
    from oracle import kde_oracle as ko

    s = np.load(t["path"], mmap_mode="r")
    w = None if t["wpath"] is None else np.load(t["wpath"])
    j = t["j"]
    name = t["names"][j]
    orc = ko.OracleSamples(np.array(s[:, [j]]), w, names=[name], ranges={k: v for k, v in t["ranges"].items() if k == name})
    t0 = time.perf_counter()
    orc.init_param(0)
    neff = orc.neff_1d(0)
    return dict(j=j, neff=float(neff), seconds=time.perf_counter() - t0)


def pair_task(t):
    """One pair by the oracle (its parameters' N_eff preset, as inside a triangle) against the GPU grid of the same pair."""
    _quiet()
    from oracle import kde_oracle as ko

    s = np.load(t["path"], mmap_mode="r")
    w = None if t["wpath"] is None else np.load(t["wpath"])
    a, b = t["pair"]
    sub = [t["names"][a], t["names"][b]]
    orc = ko.OracleSamples(np.array(s[:, [a, b]]), w, names=sub, ranges={k: v for k, v in t["ranges"].items() if k in sub})
    for k, j in enumerate((a, b)):
        orc.init_param(k)
        orc.pars[k].N_eff_kde = t["neff"][j]
    tr = {}
    t0 = time.perf_counter()
    o = orc.density_2d(0, 1, trace=tr)
    seconds = time.perf_counter() - t0
    P = o["P"]
    F = int(P.shape[0])
    row = dict(pair=[a, b], F=F, branch=tr.get("branch"), tnc="p_13" in tr, seconds=seconds, shape_ok=bool(F == t["gpu_F"]),
               bounded=int(bool(orc.pars[0].has_limits)) + int(bool(orc.pars[1].has_limits)))
    if not row["shape_ok"]:
        row["err"] = float("inf")
        return row
    G = np.asarray(np.load(t["gpu_path"], mmap_mode="r")[t["gpu_off"]:t["gpu_off"] + F * F]).reshape(F, F)
    row["err"] = float(np.max(np.abs(G - P)))
    row["sum_rel"] = float(abs(np.sum(G) - np.sum(P)) / np.sum(P))
    bw = np.array([tr.get("hx"), tr.get("hy"), tr.get("c")], dtype=float)
    gbw = np.asarray(t["gpu_bw"], dtype=float)
    row["bandwidth_rel_err"] = float(np.max(np.abs(gbw - bw)) / max(abs(bw[0]), abs(bw[1])))
    kopt = t["gpu_kopt"]
    if kopt is not None and "t_star" in tr:
        row["t_star_rel_err"] = float(abs(kopt[0] - tr["t_star"]) / abs(tr["t_star"]))
    if row["err"] > 1e-6 and row["tnc"] and kopt is not None:
        psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
        trip = np.asarray(kopt[8:11], dtype=float)
        ensembles = ko.get_h_ensembles(psi, tr["opt_N"], tr["opt_corr"])
        verdict = ko.judge_triple(trip, psi, tr["opt_N"], ensembles=ensembles)
        nscales = ko.ENSEMBLE_SCALES.index(verdict["scale"]) + 1
        ens = np.concatenate([ensembles[0]] + [e[1:] for e in ensembles[1:nscales]])  # what the verdict was reached on
        row["oracle_moves_by"] = verdict["moved"]
        row["inside_oracle_spread"], row["excess_over_spread"] = verdict["inside"], verdict["excess"]
        row["within_spread_slack_0p25"] = verdict["within_slack"]
        row["amise_ok"], row["amise_excess"], row["amise_range"] = verdict["amise_ok"], verdict["amise_excess"], verdict["amise_range"]
        row["ensemble_perturbation"], row["ensemble_members"] = verdict["scale"], verdict["members"]
        scale = np.array([np.max(np.abs(ens[:, 0])), np.max(np.abs(ens[:, 1])), 1.0])
        d = np.max(np.abs(ens - trip) / scale, axis=1)
        row["nearest_member_distance"] = float(np.min(d))  # largest relative component difference to the closest member
        row["ensemble_diameter"] = float(np.max(np.max(np.abs(ens[:, None, :] - ens[None, :, :]) / scale, axis=2)))
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsamples", type=int, default=10_000_000)
    ap.add_argument("--nparams", type=int, default=50)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--weighted", action="store_true", help="real weights w ~ Exp(1) (the block recipe's weighted variant)")
    ap.add_argument("--max-pairs", type=int, default=0, help="testing: only the first K pairs")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "parity_census_1e7.json"))
    args = ap.parse_args()
    _quiet()
    from getdist_amd import synth
    from getdist_amd.mcsamples import MCSamples

    if args.weighted:
        s, w, names, ranges = synth.block_recipe(args.nparams, args.nsamples, weighted=True, stream=1)
    else:
        s, w, names, ranges = synth.config_c3(args.nsamples, args.nparams)
    pairs = synth.triangle_pairs(args.nparams)
    if args.max_pairs:
        pairs = pairs[:args.max_pairs]
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    t0 = time.perf_counter()
    dens = mc.get2DDensities(pairs)
    dens[-1].P
    gpu_s = time.perf_counter() - t0
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="gdamd_census_", dir=shm)
    path = os.path.join(tmp, "samples.npy")
    np.save(path, np.asfortranarray(s))
    wpath = None
    if w is not None:
        wpath = os.path.join(tmp, "weights.npy")
        np.save(wpath, np.asarray(w))
    sizes = np.array([d.P.size for d in dens], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    flat = np.empty(int(offs[-1]))
    for d, o in zip(dens, offs[:-1]):
        flat[o:o + d.P.size] = d.P.ravel()
    gpu_path = os.path.join(tmp, "gpu.npy")
    np.save(gpu_path, flat)
    del flat
    gpu_neff = [float(p.N_eff_kde) for p in mc.paramNames.names]
    cores = os.cpu_count() or 1
    workers = args.workers or max(1, min(cores - 2, 96))
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    ctx = mp.get_context("spawn")
    base = dict(path=path, wpath=wpath, names=list(names), ranges=dict(ranges))
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        used = sorted({j for p in pairs for j in p})
        preps = pool.map(prep_task, [dict(base, j=j) for j in used], chunksize=1)
        neff = {r["j"]: r["neff"] for r in preps}
        neff_rel = max(abs(neff[j] - gpu_neff[j]) / neff[j] for j in used)
        tasks = [dict(base, pair=list(pr), neff=neff, gpu_path=gpu_path, gpu_off=int(offs[k]), gpu_F=int(dens[k].P.shape[0]),
                      gpu_bw=list(dens[k].bandwidth), gpu_kopt=None if dens[k].kopt is None else np.asarray(dens[k].kopt))
                 for k, pr in enumerate(pairs)]
        rows = pool.map(pair_task, tasks, chunksize=1)
    cpu_wall = time.perf_counter() - t0
    import shutil

    shutil.rmtree(tmp, ignore_errors=True)
    census = {}
    for r in rows:
        key = "%s/%d/%d" % (r["branch"], r["bounded"], r["F"])
        c = census.setdefault(key, dict(pairs=0, errs=[], loose=0, tnc=0))
        c["pairs"] += 1
        c["errs"].append(r["err"])
        c["loose"] += r["err"] > 1e-6
        c["tnc"] += bool(r["tnc"])
    loose = [r for r in rows if r["err"] > 1e-6]
    errs = np.array([r["err"] for r in rows])
    out = dict(
        config="C3 block recipe%s: %d parameters, %d pairs, N = %d; GPU through MCSamples.get2DDensities (native batch entry), "
               "oracle = oracle/kde_oracle.py in a pool of %d single-threaded workers" % (" (weighted, w ~ Exp(1))" if args.weighted else "",
                                                                                          args.nparams, len(pairs), args.nsamples, workers),
        gate="max|dP| of the max-normalised grids, GPU vs oracle, every pair",
        pairs_compared=len(rows), grid_shapes_equal=int(sum(r["shape_ok"] for r in rows)),
        pairs_within_1e_6=int(np.sum(errs <= 1e-6)), pairs_within_1e_9=int(np.sum(errs <= 1e-9)),
        pairs_above_1e_6=len(loose), worst_abs_dP=float(errs.max()), median_abs_dP=float(np.median(errs)),
        neff_max_rel_err=float(neff_rel),
        t_star_max_rel_err=float(max([r.get("t_star_rel_err", 0.0) for r in rows])),
        loose_pairs_that_use_tnc=int(sum(bool(r["tnc"]) for r in loose)),
        loose_pairs_chaotic_in_the_oracle=int(sum(r.get("oracle_moves_by", 0.0) > 1e-6 for r in loose)),
        loose_pairs_inside_the_oracle_spread=int(sum(bool(r.get("inside_oracle_spread")) for r in loose)),
        loose_pairs_inside_spread_or_as_good_in_amise=int(sum(bool(r.get("within_spread_slack_0p25") or r.get("amise_ok")) for r in loose)),
        per_class={k: dict(pairs=v["pairs"], tnc_pairs=v["tnc"], above_1e_6=int(v["loose"]), max_abs_dP=float(np.max(v["errs"])),
                           median_abs_dP=float(np.median(v["errs"])),
                           quantiles_abs_dP={q: float(np.quantile(v["errs"], float(q))) for q in ("0.5", "0.9", "0.99")})
                   for k, v in sorted(census.items())},
        loose_pairs=[dict(pair=[names[r["pair"][0]], names[r["pair"][1]]], klass="%s/%d/%d" % (r["branch"], r["bounded"], r["F"]),
                          max_abs_dP=r["err"], bandwidth_rel_err=r.get("bandwidth_rel_err"),
                          oracle_moves_by=r.get("oracle_moves_by"), inside_oracle_spread=r.get("inside_oracle_spread"),
                          excess_over_spread=r.get("excess_over_spread"), nearest_member_distance=r.get("nearest_member_distance"),
                          ensemble_diameter=r.get("ensemble_diameter"), as_good_in_amise=r.get("amise_ok"),
                          amise_excess=r.get("amise_excess")) for r in sorted(loose, key=lambda r: -r["err"])],
        seconds=dict(gpu_triangle_cold=round(gpu_s, 3), cpu_pool_wall=round(cpu_wall, 1),
                     cpu_core_seconds_pairs=round(sum(r["seconds"] for r in rows), 1),
                     cpu_core_seconds_preps=round(sum(r["seconds"] for r in preps), 1), workers=workers, host_cores=cores))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("per_class", "loose_pairs")}, indent=1))
    for k, v in out["per_class"].items():
        print(k, v)
    for r in out["loose_pairs"]:
        print(r)
    mc.ctx.close()


if __name__ == "__main__":
    main()
