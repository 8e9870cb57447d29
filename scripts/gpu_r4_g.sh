#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out/r4g
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4g/trace -o bench -- python $R/bench.py --workload config1 --steps 10 --warmup 3 --no-cpu-baseline --finishers 8 --pipeline-depth 8 > $R/gpurun_out/r4g/bench_traced.json 2> $R/gpurun_out/r4g/bench_traced.err
cd $R
DB=$(find gpurun_out/r4g/trace -name "*.db" | head -1)
python scripts/rocprof_busy.py $DB 0.5 > gpurun_out/r4g/busy.txt
cat gpurun_out/r4g/busy.txt
python scripts/rocprof_batch_timeline.py $DB -3 > gpurun_out/r4g/timeline.txt 2>&1
head -5 gpurun_out/r4g/timeline.txt
python - <<'PY'
import json
for line in open("gpurun_out/r4g/bench_traced.json"):
    if line.startswith('{'):
        j=json.loads(line); print("traced GCUPS", j['value'], j['ms_per_step'])
PY
rm -rf gpurun_out/r4g/trace
