"""
Summaries of round 4's separate `rocprofv3 --kernel-trace --pmc <counter>` passes (scripts/r04_evidence_run.sh):
  <dir>/pmc_hist_{FETCH_SIZE,WRITE_SIZE,SQ}   scripts/pmc_hist2d.py      -> r04_pmc_hist2d.json (+ the file bench.py falls back to)
  <dir>/pmc_kopt_{FETCH_SIZE,WRITE_SIZE}      scripts/tune_kopt.py       -> r04_pmc_kopt.json
  <dir>/pmc_{c2,c4,c5}_{FETCH_SIZE,WRITE_SIZE} scripts/run_configs.py cN -> r04_pmc_configs.json (top kernels of each config)
HBM bytes per launch = FETCH_SIZE (KiB) x 2 (gfx950 tallies 128-byte requests at 64 B for 16-byte-per-lane streams,
MI355X_MICROARCH.md) x 1024 + WRITE_SIZE (KiB) x 1024, averaged over a kernel's launches.
Usage: python scripts/summarise_pmc_r04.py <dir>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

D = sys.argv[1]


def read(tag):
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(D, tag, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    return per, dur


def traffic(prefix, keep=None, top=8):
    fetch, dur = read(prefix + "_FETCH_SIZE")
    write, _ = read(prefix + "_WRITE_SIZE")
    out = {}
    for name in fetch:
        if keep and not any(k in name for k in keep):
            continue
        fs = fetch[name]["FETCH_SIZE"]
        ws = write.get(name, {}).get("WRITE_SIZE", [0.0])
        byt = (2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024
        ms = sum(dur[name]) / len(dur[name])
        out[name] = dict(launches=len(fs), fetch_KiB=sum(fs) / len(fs), write_KiB=sum(ws) / len(ws), hbm_bytes_per_launch=byt,
                         ms_under_the_profiler=ms, total_ms=sum(dur[name]), TBps=byt / ms / 1e9 if ms > 0 else None,
                         frac_of_8TBps=byt / ms / 1e9 / 8.0 if ms > 0 else None)
    if not keep:  # the heaviest kernels only
        out = dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"])[:top])
    return out


def main():
    res = {}
    hist = traffic("pmc_hist", keep=["hist2d"])
    sq, _ = read("pmc_hist_SQ")
    for name, ent in hist.items():
        for counter, vals in sq.get(name, {}).items():
            ent[counter] = sum(vals) / len(vals)
        if ent.get("SQ_LDS_IDX_ACTIVE"):
            ent["lds_conflict_fraction"] = ent["SQ_LDS_BANK_CONFLICT"] / ent["SQ_LDS_IDX_ACTIVE"]
    if hist:
        main_kernel = [k for k in hist if k.startswith("k_hist2d_u8")][0]
        top = dict(N=10_000_000, n=50, F=256, weighted=False, pairs=1200, kernel=main_kernel,
                   hbm_bytes_per_launch=hist[main_kernel]["hbm_bytes_per_launch"], kernels=hist,
                   source="rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_*} -- python scripts/pmc_hist2d.py (separate passes)")
        json.dump(top, open(os.path.join(D, "r04_pmc_hist2d.json"), "w"), indent=1)
        res["hist2d"] = {k: round(v["frac_of_8TBps"], 3) for k, v in hist.items()}
    kopt = traffic("pmc_kopt", keep=["k_kopt2d", "k_dct_pass", "k_get_h", "k_gemm_nt"])
    if kopt:
        json.dump(dict(kernels=kopt, source="the same two passes over python scripts/tune_kopt.py (1200 pairs through gd_kopt2d)"),
                  open(os.path.join(D, "r04_pmc_kopt.json"), "w"), indent=1)
        res["kopt"] = {k: (round(v["hbm_bytes_per_launch"] / 1e9, 2), round(v["ms_under_the_profiler"], 3)) for k, v in kopt.items()}
    cfg = {}
    for c in ("c2", "c4", "c5"):
        t = traffic("pmc_" + c)
        if t:
            cfg[c] = t
    if cfg:
        json.dump(cfg, open(os.path.join(D, "r04_pmc_configs.json"), "w"), indent=1)
        res["configs"] = {c: {k: (round(v["hbm_bytes_per_launch"] / 1e9, 3), round(v["ms_under_the_profiler"], 3), round(v["frac_of_8TBps"] or 0, 3))
                              for k, v in t.items()} for c, t in cfg.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
