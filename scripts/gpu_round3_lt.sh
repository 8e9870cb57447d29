#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_longtarget.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --workload nhmmer --steps 2 --warmup 0 --nhmmer-searches 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        n = json.loads(l)['nhmmer']; print(n['value'], n['s_per_search'], n['ssv_scan_kernel_ms'], n['ssv_scan_gcups'], n['ms'])"
