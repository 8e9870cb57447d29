#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_filters.py tests/test_gpu_envelopes.py -x -q 2>&1 | tail -3
python scripts/env_config1.py 2>&1 | tail -2
for i in 1 2; do python bench.py --workload config1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_query'], d['stages']['device_ms'])"; done
