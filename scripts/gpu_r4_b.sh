#!/bin/bash
# round 4: ensemble kernels after the first optimisation pass -- parity tests, kernel durations inside a short headline run
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_ensembles.py -q -x 2>&1 | tail -25 > gpurun_out/r4b/ens.txt
tail -3 gpurun_out/r4b/ens.txt
timeout 600 python bench.py --debug-option trace_finish=1 --workload config1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4b/bench_config1.txt 2> gpurun_out/r4b/bench_config1.err
tail -c 600 gpurun_out/r4b/bench_config1.txt
grep "finish\]" gpurun_out/r4b/bench_config1.err | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4b/prof -o t -- python $R/bench.py --workload config1 --steps 3 --warmup 1 --no-cpu-baseline --spinup-max 2 > $R/gpurun_out/r4b/prof.log 2>&1
cd $R
python - <<'PY'
import sqlite3,glob
dbs=glob.glob("gpurun_out/r4b/prof/**/*.db", recursive=True)
c=sqlite3.connect(dbs[0])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
kt=[t for t in tabs if t.startswith('kernels')][0]
rows=list(c.execute(f"select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from {kt} group by name order by avg(duration)*count(*) desc limit 16"))
with open("gpurun_out/r4b/kernels.txt","w") as f:
    for r in rows:
        line="%-70s n=%5d avg %9.1f us min %9.1f max %9.1f" % (r[0][:70], r[1], r[2], r[3], r[4])
        print(line); f.write(line+"\n")
PY
rm -rf gpurun_out/r4b/prof
