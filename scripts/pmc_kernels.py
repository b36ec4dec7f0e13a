"""
HBM counter traffic per kernel of any command: three separate rocprofv3 passes (FETCH_SIZE; WRITE_SIZE; the SQ group) as
MI355X_MICROARCH.md prescribes -- one counter group per pass, never with the tracing domains -- and a per-kernel summary:
  bytes per launch = FETCH_SIZE (KiB) x 2 (gfx950 tallies 128-byte requests at 64 B for 16-byte-per-lane streams) x 1024
                     + WRITE_SIZE (KiB) x 1024,
  the average duration under the profiler, bytes / duration against the 8 TB/s peak, VALU instructions per LDS instruction
  and the LDS bank-conflict fraction.
Usage (GPU box):  python scripts/pmc_kernels.py <out.json> [--match substr,substr] -- <command ...>
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

HBM_PEAK = 8.0e12


def one_pass(cmd, counters, tmp):
    out = os.path.join(tmp, counters[0])
    full = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", "p", "--"] + cmd
    r = subprocess.run(full, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 pass %s failed:\n%s" % (counters, r.stdout.decode()[-2000:]))
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[name][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    return per, {k: list(v.values()) for k, v in dur.items()}


def main():
    args = sys.argv[1:]
    sep = args.index("--")
    out_path = args[0]
    match = None
    if "--match" in args[:sep]:
        match = args[args.index("--match") + 1].split(",")
    cmd = args[sep + 1:]
    tmp = tempfile.mkdtemp(prefix="gdamd_pmc_", dir="/tmp")
    try:
        fetch, dur = one_pass(cmd, ["FETCH_SIZE"], tmp)
        write, _ = one_pass(cmd, ["WRITE_SIZE"], tmp)
        sq, _ = one_pass(cmd, ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"], tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for name in sorted(fetch, key=lambda k: -sum(dur.get(k, [0]))):
        if match and not any(m in name for m in match):
            continue
        fs, ws = fetch[name]["FETCH_SIZE"], write.get(name, {}).get("WRITE_SIZE", [0.0])
        ms = sum(dur[name]) / len(dur[name])
        nbytes = (2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024
        ent = dict(launches=len(fs), ms_per_launch_under_the_profiler=round(ms, 4), fetch_KiB=round(sum(fs) / len(fs), 1),
                   write_KiB=round(sum(ws) / len(ws), 1), hbm_bytes_per_launch=nbytes,
                   frac_of_hbm_peak=round(nbytes / (ms * 1e-3) / HBM_PEAK, 3) if ms > 0 else None)
        s = sq.get(name, {})
        if s.get("SQ_INSTS_LDS") and sum(s["SQ_INSTS_LDS"]) > 0:
            ent["valu_per_lds_instruction"] = round(sum(s["SQ_INSTS_VALU"]) / sum(s["SQ_INSTS_LDS"]), 2)
        if s.get("SQ_LDS_IDX_ACTIVE") and sum(s["SQ_LDS_IDX_ACTIVE"]) > 0:
            ent["lds_conflict_fraction"] = round(sum(s["SQ_LDS_BANK_CONFLICT"]) / sum(s["SQ_LDS_IDX_ACTIVE"]), 3)
        res[name] = ent
    top = dict(command=" ".join(cmd), source="rocprofv3 --kernel-trace --pmc, three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*); "
               "bytes = FETCH_SIZE KiB x 2 x 1024 + WRITE_SIZE KiB x 1024 (MI355X_MICROARCH.md)", hbm_peak=HBM_PEAK, kernels=res)
    json.dump(top, open(out_path, "w"), indent=1)
    for k, v in list(res.items())[:14]:
        print("%-46s %7.1f us  %8.1f MB  %.2f of HBM" % (k[:46], v["ms_per_launch_under_the_profiler"] * 1e3, v["hbm_bytes_per_launch"] / 1e6,
                                                          v["frac_of_hbm_peak"] or 0))


if __name__ == "__main__":
    main()
