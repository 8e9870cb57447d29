#!/usr/bin/env python
"""Export the kernel statistics of a rocprofv3 (rocpd sqlite) run as a small markdown/CSV summary.

usage: rocprof_summary.py <results.db> <out.md> [title]
The .db comes from:  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d <dir> -o bench -- python bench.py ...
"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else db
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
detail = {}
for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, "
                   "min(duration), max(duration), avg(duration), count(*) from kernels group by name"):
    detail[r[0]] = r[1:]
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds (rocpd `top_kernels` / `kernels` views).\n\n")
    f.write("| kernel | calls | total us | avg us | % | grid | wg | LDS B | VGPR | AGPR | SGPR | scratch | min us | max us |\n")
    f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    for name, calls, tot, avg, pct in rows:
        d = detail.get(name, [None] * 11)
        short = name if len(name) < 90 else name[:87] + "..."
        f.write(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.1f} | {d[0]} | {d[1]} | {d[2]} | {d[3]} | {d[4]} | {d[5]} | {d[6]} | "
                f"{(d[7] or 0) / 1e3:.2f} | {(d[8] or 0) / 1e3:.2f} |\n")
print(open(out).read())
