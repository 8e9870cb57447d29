#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_envelopes.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --workload config1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_query'], d['stages']['device_ms'])"
cp pyhmmer_amd/libp7x.so /tmp/libp7x_keep.so
scripts/env_variant.sh scratch_variants/env_prof.o -- python scripts/env_phase_profile.py 2>&1 | tail -5
cp /tmp/libp7x_keep.so pyhmmer_amd/libp7x.so
