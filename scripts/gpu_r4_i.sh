#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out/r4i
cd $R
timeout 600 python -m pytest tests/test_gpu_longtarget.py -q -x -k "stream_of_queries" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4i/trace -o scan -- python $R/bench.py --workload scan --no-cpu-baseline --steps 2 --warmup 1 --spinup-max 1 > $R/gpurun_out/r4i/scan_traced.json 2> $R/gpurun_out/r4i/scan_traced.err
cd $R
DB=$(find gpurun_out/r4i/trace -name "*.db" | head -1)
python - $DB <<'PY'
import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name, start, end from kernels order by start"))
# the scan is the last stretch: take kernels after the last gap > 200 ms
cut=0
for i in range(1,len(rows)):
    if rows[i][1]-max(r[2] for r in rows[max(0,i-50):i])>2e8: cut=i
rows=rows[cut:]
t0,t1=rows[0][1],max(r[2] for r in rows)
print(f"{len(rows)} launches over {(t1-t0)/1e6:.1f} ms")
tot={}
for n,s,e in rows:
    k=n.replace("void p7x::","").replace("p7x::","").split("(")[0][:44]
    a=tot.setdefault(k,[0,0]); a[0]+=1; a[1]+=e-s
ev=sorted([(r[1],1) for r in rows]+[(r[2],-1) for r in rows]); d=0; last=t0; busy=0
for t,x in ev:
    if d>0: busy+=t-last
    d+=x; last=t
print(f"device busy {busy/1e6:.1f} ms; sum of durations {sum(v[1] for v in tot.values())/1e6:.1f} ms")
for k,(n,dd) in sorted(tot.items(), key=lambda kv:-kv[1][1])[:18]:
    print(f"  {k:44s} n={n:5d} total {dd/1e6:8.2f} ms avg {dd/n/1e3:9.1f} us")
PY
python - <<'PY'
import json
for line in open("gpurun_out/r4i/scan_traced.json"):
    if line.startswith('{'):
        j=json.loads(line); print("scan", j.get('scan'))
PY
rm -rf gpurun_out/r4i/trace
