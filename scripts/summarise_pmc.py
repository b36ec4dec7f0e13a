"""
Turn the rocprofv3 --pmc CSVs of scripts/pmc_hist2d.py into profiles/r03_pmc_hist2d.json (read by bench.py's roofline
block).  Usage: python scripts/summarise_pmc.py <fetch.csv> <write.csv> [<sq.csv>]
HBM bytes per launch = FETCH_SIZE (KiB) x 2 (gfx950 tallies 128-B requests at 64 B for 16-B/lane streams,
MI355X_MICROARCH.md HBM section) x 1024 + WRITE_SIZE (KiB) x 1024, averaged over the launches of each kernel.
"""
import csv
import json
import os
import sys
from collections import defaultdict


def read(path):
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    return per, dur


def main():
    fetch, dur = read(sys.argv[1])
    write, _ = read(sys.argv[2])
    sq = read(sys.argv[3])[0] if len(sys.argv) > 3 else {}
    out = {}
    for name in fetch:
        if "hist2d" not in name:
            continue
        fs = fetch[name]["FETCH_SIZE"]
        ws = write.get(name, {}).get("WRITE_SIZE", [0.0])
        ent = dict(launches=len(fs), fetch_KiB=sum(fs) / len(fs), write_KiB=sum(ws) / len(ws),
                   hbm_bytes_per_launch=(2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024,
                   ms_under_the_profiler=sum(dur[name]) / len(dur[name]))
        for counter, vals in sq.get(name, {}).items():
            ent[counter] = sum(vals) / len(vals)
        if "SQ_LDS_BANK_CONFLICT" in ent and ent.get("SQ_LDS_IDX_ACTIVE"):
            ent["lds_conflict_fraction"] = ent["SQ_LDS_BANK_CONFLICT"] / ent["SQ_LDS_IDX_ACTIVE"]
        out[name] = ent
    main_kernel = [k for k in out if k.startswith("k_hist2d_u8")][0]  # k_hist2d_u8_pf<3> since round 3
    top = dict(N=10_000_000, n=50, F=256, weighted=False, pairs=1200, kernel=main_kernel,
               hbm_bytes_per_launch=out[main_kernel]["hbm_bytes_per_launch"], kernels=out,
               source="rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_*} -- python scripts/pmc_hist2d.py (separate passes)")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_pmc_hist2d.json")
    json.dump(top, open(path, "w"), indent=1)
    print(json.dumps({k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
