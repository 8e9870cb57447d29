#!/usr/bin/env python
"""Text timeline of one batch from a rocprofv3 kernel trace (rocpd sqlite): every kernel launch between two consecutive
batch boundaries (a boundary = a gap with no kernel running), with its stream, start, duration and grid.
usage: rocprof_batch_timeline.py <results.db> [which_batch]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(c.execute(f"select name, start, end, {qcol}, grid_x, grid_y from kernels order by start"))
# split into batches at idle gaps > 1 ms
batches, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[1] - last_end > 1e6 and cur:
        batches.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, r[2])
if cur:
    batches.append(cur)
b = batches[which]
t0 = b[0][1]
print(f"{len(batches)} batches; batch {which}: {len(b)} launches, {(max(r[2] for r in b) - t0) / 1e6:.3f} ms from first start to last end")
busy = sorted([(r[1], 1) for r in b] + [(r[2], -1) for r in b])
depth, last, tot = 0, t0, 0
for t, d in busy:
    if depth > 0:
        tot += t - last
    depth += d; last = t
print(f"device busy {tot / 1e6:.3f} ms; sum of kernel durations {sum(r[2] - r[1] for r in b) / 1e6:.3f} ms")
for r in b:
    n = r[0].replace("void p7x::", "").replace("p7x::", "").split("(")[0][:40]
    print(f"  {(r[1] - t0) / 1e6:8.3f} +{(r[2] - r[1]) / 1e6:7.3f}  q{r[3]}  grid {r[4]}x{r[5]}  {n}")
