cd /tmp && export TMPDIR=/tmp
ENV_PLANTED=0.6 timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o t -- python $GRAFT_REPO_ROOT/scripts/env_by_length.py 60 100 128 > $GRAFT_REPO_ROOT/gpurun_out/$1.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("gpurun_out/$1/*.db")[0])
for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3 from kernels where name like '%env_kernel%' group by name"):
    print("%-50s n=%3d avg %9.1f us min %9.1f us" % (r[0][:50], r[1], r[2], r[3]))
PY
