#!/bin/bash
# kernel durations of the envelope kernel (rocprofv3 kernel trace) on the headline workload, per variant object
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
i=0
for o in $1; do
  i=$((i+1))
  objs=$(ls pyhmmer_amd/csrc/build/*.o | grep -v p7x_envelope.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pyhmmer_amd/libp7x.so $objs $o -lpthread
  (cd /tmp && export TMPDIR=/tmp && DEPTH=${DEPTH:-0} timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/kt$i -o t -- python $R/scripts/env_config1.py > $R/gpurun_out/kt$i.log 2>&1)
  python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("gpurun_out/kt$i/*.db")[0])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt=[t for t in tabs if t=='kernels'] and 'kernels'
for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels where name like '%env_kernel%' group by name"):
    print("$o %-40s n=%3d avg %9.1f us min %9.1f max %9.1f" % (r[0][:40], r[1], r[2], r[3], r[4]))
PY
done
