#!/usr/bin/env python
"""Dump the kernel records of a rocprofv3 kernel trace (rocpd sqlite) as CSV -- name, start and end in microseconds from the
first record, queue, grid, workgroup, LDS bytes -- for the launches that start in the last <seconds> of the trace (0: all).
usage: rocprof_dump_kernels.py <results.db> <out.csv> [seconds]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
last = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(c.execute(f"select name, start, end, {qcol}, grid_x, grid_y, workgroup_x, lds_size from kernels order by start"))
t_end = max(r[2] for r in rows)
if last > 0:
    rows = [r for r in rows if r[1] >= t_end - last * 1e9]
t0 = rows[0][1]
with open(sys.argv[2], "w") as f:
    f.write("name,start_us,end_us,queue,grid_x,grid_y,wg,lds\n")
    for n, s, e, q, gx, gy, wg, lds in rows:
        k = n.replace("void p7x::", "").replace("p7x::", "").split("(")[0]
        f.write(f"\"{k}\",{(s - t0) / 1e3:.1f},{(e - t0) / 1e3:.1f},{q},{gx},{gy},{wg},{lds}\n")
print(f"{len(rows)} kernel records, {(rows[-1][2] - t0) / 1e6:.1f} ms")
