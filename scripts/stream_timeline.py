"""Per-stream timeline of one bench step from a rocprofv3 kernel trace (CSV): the step starts at a k_col_pass1 launch
(the base statistics); consecutive launches of one kernel on one stream are merged.  Usage:
    python scripts/stream_timeline.py <kernel_trace.csv> [step index, default -2] [min ms to print, default 0.05]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    r["q"] = r.get("Stream_Id") or r.get("Queue_Id")
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["n"].startswith("k_col_pass1") or r["n"].startswith("k_col_shift")]
# (a step of a multi-rank share has one statistics pass as well; launches of it closer than 2 ms belong to one step)
starts = [i for k, i in enumerate(starts) if k == 0 or rows[i]["s"] - rows[starts[k - 1]]["s"] > 2_000_000]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
a = starts[which]
b = starts[which + 1] if which + 1 != 0 and which + 1 < len(starts) else len(rows)
step = rows[a:b]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
busy, depth, last = 0, 0, t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print("step span %.2f ms, GPU busy (union) %.2f ms, sum of kernel durations %.2f ms, %d launches"
      % ((t1 - t0) / 1e6, busy / 1e6, sum(r["e"] - r["s"] for r in step) / 1e6, len(step)))
streams = collections.OrderedDict()
for r in step:
    streams.setdefault(r["q"], []).append(r)
for q, rs in streams.items():
    print("---- stream %s: %d launches, %.2f ms of kernels" % (q, len(rs), sum(r["e"] - r["s"] for r in rs) / 1e6))
    merged = []
    for r in rs:
        if merged and merged[-1][0] == r["n"] and r["s"] - merged[-1][2] < 30000:
            merged[-1][2] = r["e"]
            merged[-1][3] += 1
            merged[-1][4] += r["e"] - r["s"]
        else:
            merged.append([r["n"], r["s"], r["e"], 1, r["e"] - r["s"]])
    prev = None
    for n, s, e, c, d in merged:
        gap = "" if prev is None or s - prev < 30000 else "   (idle %.3f)" % ((s - prev) / 1e6)
        if d / 1e6 >= floor or gap:
            print("  %8.3f -> %8.3f  %7.3f ms x%-3d %s%s" % ((s - t0) / 1e6, (e - t0) / 1e6, d / 1e6, c, n[:70], gap))
        prev = e
tot = collections.Counter()
for r in step:
    tot[r["n"]] += r["e"] - r["s"]
print("---- totals")
for n, t in tot.most_common(30):
    print("%8.3f ms  %s" % (t / 1e6, n[:90]))
