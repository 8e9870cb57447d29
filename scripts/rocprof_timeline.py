#!/usr/bin/env python
"""Device occupancy over time from a rocprofv3 kernel trace (rocpd sqlite): fraction of wall time with at least one
kernel running, time with MSV running, and a text timeline of one steady-state stretch.

usage: rocprof_timeline.py <results.db> [window_ms]
"""
import sqlite3
import sys

db = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
if not rows:
    sys.exit("no kernels")
t0 = rows[0][1]
ev = [(r[0], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, r[3] if qcol else 0) for r in rows]


def short(n):
    n = n.replace("void p7x::", "").replace("p7x::", "")
    return n.split("(")[0][:28]


# steady state: the middle third of the MSV launches
msv = [e for e in ev if "msv_fast_kernel" in e[0]]
a, b = msv[len(msv) // 3][1], msv[2 * len(msv) // 3][1]
inwin = [e for e in ev if e[2] > a and e[1] < b]
pts = sorted([(max(e[1], a), 1) for e in inwin] + [(min(e[2], b), -1) for e in inwin])
busy = 0.0; depth = 0; last = a; hist = {}
for t, d in pts:
    if depth > 0:
        busy += t - last
    hist[depth] = hist.get(depth, 0.0) + (t - last)
    depth += d; last = t
nq = len([e for e in msv if a <= e[1] < b])
print(f"steady window {b - a:.1f} ms, {nq} queries -> {(b - a) / max(nq, 1):.2f} ms/query; device busy {100 * busy / (b - a):.1f} %")
print("time share by number of concurrently running kernels:", {k: f"{100 * v / (b - a):.1f}%" for k, v in sorted(hist.items())})
msv_busy = sum(min(e[2], b) - max(e[1], a) for e in inwin if "msv_fast" in e[0])
print(f"MSV running {100 * msv_busy / (b - a):.1f} % of the window")
per = {}
for e in inwin:
    per[short(e[0])] = per.get(short(e[0]), 0.0) + (min(e[2], b) - max(e[1], a))
print("kernel-time per query (ms):", {k: round(v / max(nq, 1), 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]})
print(f"\ntimeline of {win} ms from t={a:.1f} ms (start end dur queue kernel):")
for e in ev:
    if e[1] >= a and e[1] < a + win and e[2] - e[1] > 0.02:
        print(f"  {e[1] - a:8.3f} {e[2] - a:8.3f} {e[2] - e[1]:7.3f}  q{e[3]}  {short(e[0])}")
