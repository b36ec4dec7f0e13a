"""One delivered triangle on the time axis: kernels by stream (first / last launch of the pipeline's stages) and the result copies
(device -> host) from a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py.
    python scripts/latency_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [step index, default -2]"""
import csv
import glob
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
rows = list(csv.DictReader(open(kt)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
rows.sort(key=lambda r: r["s"])
starts = [r["s"] for r in rows if r["n"].startswith("k_col_shift") or r["n"].startswith("k_col_pass1")]
starts = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 2_000_000]
t0 = starts[which]
t1 = starts[which + 1] if which + 1 != 0 and which + 1 < len(starts) else rows[-1]["e"] + 1
step = [r for r in rows if t0 <= r["s"] < t1 and r["s"] - t0 < 60_000_000]
print("step of %d launches, last kernel ends at %.2f ms; step starts (ms, relative): %s"
      % (len(step), (max(r["e"] for r in step) - t0) / 1e6, [round((x - t0) / 1e6, 2) for x in starts]))
groups = [("statistics", ("k_col_shift", "k_cov_")), ("quantile select", ("k_qlin", "k_qsel")), ("N_eff lag sums", ("k_kde_lag",)),
          ("pre-binning", ("k_bucket_lut", "k_prebin")), ("main binning", ("k_hist2d_u8",)), ("shear chain", ("k_minmax_affine", "k_hist2d_f64")),
          ("up-scaled classes", ("k_hist2d_u16",)), ("optimiser stage A", ("k_dct_pass", "k_kopt2d")), ("get_h (TNC)", ("k_get_h",)),
          ("convolution", ("k_rows_", "k_col_conv", "k_win_", "k_boundary", "k_normalise", "k_window_sat"))]
for name, pre in groups:
    rs = [r for r in step if r["n"].startswith(pre)]
    if rs:
        print("  %-20s %7.2f -> %7.2f ms   (%3d launches, %6.2f ms of kernel time)"
              % (name, (min(r["s"] for r in rs) - t0) / 1e6, (max(r["e"] for r in rs) - t0) / 1e6, len(rs), sum(r["e"] - r["s"] for r in rs) / 1e6))
if mt:
    cp = list(csv.DictReader(open(mt[0])))
    for c in cp:
        c["s"], c["e"] = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
    end = (starts[which + 1] + 15_000_000) if which + 1 != 0 and which + 1 < len(starts) else 1 << 62
    mine = [c for c in cp if t0 <= c["s"] < end and "DEVICE_TO_HOST" in c["Direction"] and c["e"] - c["s"] > 100_000]  # (the trace has no sizes)
    if mine:
        busy = sum(c["e"] - c["s"] for c in mine)
        print("  result copies (device -> host, longer than 0.1 ms): %d copies; first starts %.2f ms, last ends %.2f ms; copy-engine time "
              "%.2f ms" % (len(mine), (min(c["s"] for c in mine) - t0) / 1e6, (max(c["e"] for c in mine) - t0) / 1e6, busy / 1e6))
        for c in mine[:60]:
            print("      stream %-3s %7.2f -> %7.2f  (%.2f ms)" % (c["Stream_Id"], (c["s"] - t0) / 1e6, (c["e"] - t0) / 1e6, (c["e"] - c["s"]) / 1e6))
