#!/bin/bash
# kernel durations of a command with nothing else in flight: scripts/kernel_times.sh <outdir> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT -o t -- "$@" > $R/$OUT.log 2>&1
cd $R
python - <<PY
import sqlite3,glob
db=glob.glob("$OUT/*.db")[0]
c=sqlite3.connect(db)
for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, grid_x, grid_y from kernels group by name order by avg(duration)*count(*) desc limit 14"):
    print("%-60s n=%4d avg %9.1f us min %9.1f max %9.1f grid %s x %s" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
