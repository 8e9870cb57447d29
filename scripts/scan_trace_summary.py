#!/usr/bin/env python
"""Markdown summary of a kernel trace of the scan orientation (scripts/rocprof_dump_kernels.py CSV of scripts/config3_scan.py 20000 trace):
the last pass of the trace -- kernel families (launches, summed / median / longest duration), launches in flight over time.
usage: scan_trace_summary.py <kernels.csv> <out.md> <title>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = float(r["start_us"]) / 1e3, float(r["end_us"]) / 1e3
rows.sort(key=lambda r: r["s"])
# the last pass: from the last idle gap of more than 21 ms that is followed by a few hundred launches, to the end of the trace
end, cuts = 0.0, [0]
for i, r in enumerate(rows):
    if i and r["s"] - end > 21.0:
        cuts.append(i)
    end = max(end, r["e"])
cuts.append(len(rows))
cut = max([a for a, b in zip(cuts[:-1], cuts[1:]) if b - a >= 200] or [0])
last = rows[cut:]
t0, t1 = last[0]["s"], max(r["e"] for r in last)
fam = collections.defaultdict(list)
for r in last:
    fam[re.sub(r"<.*", "", r["name"])].append(r["e"] - r["s"])
total = sum(sum(v) for v in fam.values())
with open(sys.argv[2], "w") as f:
    f.write(f"# {sys.argv[3]}\n\nLast pass of the trace: {len(last)} launches over {t1 - t0:.0f} ms, {total:.0f} ms of kernel time "
            f"({total / (t1 - t0):.1f} launches in flight on average).\n\n| kernel | launches | sum ms | median ms | max ms |\n|---|---:|---:|---:|---:|\n")
    for k, v in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        f.write(f"| `{k}` | {len(v)} | {sum(v):.1f} | {v[len(v) // 2]:.2f} | {v[-1]:.2f} |\n")
    f.write("\nLaunches in flight, mean per 10 ms of the pass:\n\n`")
    bins = collections.Counter()
    for r in last:
        i = int((r["s"] - t0) // 10)
        while t0 + i * 10 < r["e"]:
            bins[i] += min(r["e"], t0 + i * 10 + 10) - max(r["s"], t0 + i * 10)
            i += 1
    f.write(" ".join(f"{bins[i] / 10:.1f}" for i in range(int((t1 - t0) // 10) + 1)) + "`\n")
print(open(sys.argv[2]).read())
