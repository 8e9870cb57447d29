#!/usr/bin/env python
"""Device occupancy of a rocprofv3 kernel trace (rocpd sqlite) over its busiest stretch: wall time between the first
and the last launch of the last <frac> of the trace, time with at least one kernel running, sum of kernel durations,
and the kernels by total duration.  usage: rocprof_busy.py <results.db> [skip_first_fraction]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(c.execute("select name, start, end from kernels order by start"))
t_first, t_last = rows[0][1], max(r[2] for r in rows)
cut = t_first + skip * (t_last - t_first)
rows = [r for r in rows if r[1] >= cut]
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
depth, last, busy, weighted = 0, t0, 0, 0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d; last = t
print(f"{len(rows)} launches over {(t1 - t0) / 1e6:.2f} ms: device busy {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.0f} %), sum of kernel durations {sum(r[2] - r[1] for r in rows) / 1e6:.2f} ms")
tot = {}
for n, s, e in rows:
    k = n.replace("void p7x::", "").replace("p7x::", "").split("(")[0][:48]
    a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k:48s} n={n:5d} total {d / 1e6:9.2f} ms avg {d / n / 1e3:9.1f} us")
