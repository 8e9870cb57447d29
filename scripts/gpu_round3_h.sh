#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in 1 2; do
P7X_LT_DEBUG=1 timeout 300 python bench.py --workload nhmmer --steps 2 --warmup 0 --nhmmer-searches 3 --nhmmer-envelopes $e 2> gpurun_out/lt_env$e.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        n = json.loads(l)['nhmmer']; print('envelopes $e', n['value'], n['s_per_search'], n['ms'])"
grep "host phase" gpurun_out/lt_env$e.err | tail -9
done
