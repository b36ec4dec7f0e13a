"""Summarise a rocprofv3 kernel trace of bench.py: find the steps (each starts with k_col_pass1), print the busy /
idle split of the last one and the per-kernel totals inside it."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["n"].startswith("k_col_pass1")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
a, b = starts[which], starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
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
tot = collections.Counter()
cnt = collections.Counter()
for r in step:
    tot[r["n"]] += r["e"] - r["s"]
    cnt[r["n"]] += 1
for n, t in tot.most_common(45):
    print("%8.3f ms %4d  %s" % (t / 1e6, cnt[n], n[:90]))
if len(sys.argv) > 3:  # timeline with gaps > 50 us
    prev = t0
    for r in step:
        if r["s"] - prev > 50000:
            print("gap %.3f ms before %s at %.3f" % ((r["s"] - prev) / 1e6, r["n"][:50], (r["s"] - t0) / 1e6))
        prev = max(prev, r["e"])
