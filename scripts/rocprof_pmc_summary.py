#!/usr/bin/env python
"""Summarise PMC counters of rocprofv3 --pmc runs (rocpd sqlite) per kernel: mean value per dispatch.

usage: rocprof_pmc_summary.py <out.md> <title> <results.db> [<results.db> ...]
"""
import sqlite3
import sys

out, title, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = {}
for db in dbs:
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    for k, cn, v, n in c.execute(f"select {kcol}, {ccol}, sum({vcol}), count(distinct dispatch_id) from counters_collection group by {kcol}, {ccol}"):
        agg.setdefault(k, {})[cn] = (v / max(n, 1), n)
with open(out, "w") as f:
    f.write(f"# {title}\n\nMean counter value per dispatch (rocprofv3 --pmc, separate passes per counter group).\n\n")
    for k, d in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        f.write(f"## `{k[:100]}`\n\n| counter | mean per dispatch | dispatches |\n|---|---:|---:|\n")
        for cn, (v, n) in sorted(d.items()):
            f.write(f"| {cn} | {v:.4g} | {n} |\n")
        f.write("\n")
print(open(out).read()[:6000])
