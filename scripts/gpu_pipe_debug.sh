#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pipe
for i in 1 2 3 4 5; do
  P7X_PIPE_DEBUG=1 python bench.py --workload config1 --no-cpu-baseline > gpurun_out/pipe/run$i.json 2> gpurun_out/pipe/run$i.err
  python -c "
import json
for l in open('gpurun_out/pipe/run$i.json'):
    if l.startswith('{'):
        d = json.loads(l); print('run$i', d['value'], d['ms_per_query'])"
  tail -c 400000 gpurun_out/pipe/run$i.err > gpurun_out/pipe/run$i.tail; rm gpurun_out/pipe/run$i.err
done
