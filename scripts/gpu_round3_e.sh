#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/fin
for fin in 2 4; do
P7X_FINISH_DEBUG=1 P7X_PIPE_DEBUG=1 python bench.py --workload pfam --no-cpu-baseline --finishers $fin --pfam-profiles 5000 2> gpurun_out/fin/fin$fin.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['pfam']; print('pfam finishers $fin', p['value'], p.get('seconds'), p.get('ms_per_profile'))"
tail -c 600000 gpurun_out/fin/fin$fin.err > gpurun_out/fin/fin$fin.tail; rm gpurun_out/fin/fin$fin.err
done
