#!/bin/bash
# HBM traffic of the line's dominant kernel (the fast MSV launches: msv_tier_kernel<T> / msv_fast_kernel<R, K, half>) from
# rocprofv3 PMC passes, one counter per run as MI355X_MICROARCH.md prescribes (FETCH_SIZE, WRITE_SIZE; gfx950: 2 x FETCH_SIZE),
# next to the algorithmic bytes of the same launches: scripts/pfam_pmc.sh <outdir under gpurun_out> [profiles]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/${1:-gpurun_out/pfam_pmc}; NP=${2:-600}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $O/$c -o pmc -- python $R/scripts/pfam_phases.py $NP > $O/$c.log 2>&1
done
cd $R
python - <<PY
import sqlite3, glob, json
o = "$O"
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(f"{o}/{c}/**/*.db", recursive=True)[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    rows = list(con.execute(f"select {kcol}, sum({vcol}), count(distinct dispatch_id) from counters_collection where {ccol} = '{c}' group by {kcol}"))
    tot[c] = {k: (v, n) for k, v, n in rows}
line = [json.loads(l) for l in open(f"{o}/FETCH_SIZE.log") if l.startswith("{")][-1]
sel = lambda k: ("msv_tier_kernel" in k) or ("msv_fast_kernel" in k)
# the phases script runs the library twice (64 profiles to warm up, then all): count both in the algorithmic bytes
fetch_kb = sum(v for k, (v, n) in tot["FETCH_SIZE"].items() if sel(k)); write_kb = sum(v for k, (v, n) in tot["WRITE_SIZE"].items() if sel(k))
launches = sum(n for k, (v, n) in tot["FETCH_SIZE"].items() if sel(k))
traffic = 2.0 * fetch_kb * 1024 + write_kb * 1024
out = {"run": line, "fast_msv_launches": launches, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "traffic_bytes": traffic,
       "correction": "2 x FETCH_SIZE (gfx950: rocprofv3 tallies the 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; KB = 1024 B",
       "per_kernel": {k: {"fetch_kb": tot["FETCH_SIZE"][k][0], "write_kb": tot["WRITE_SIZE"].get(k, (0, 0))[0], "dispatches": tot["FETCH_SIZE"][k][1]} for k in tot["FETCH_SIZE"] if sel(k)}}
json.dump(out, open(f"{o}/pfam_traffic.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
