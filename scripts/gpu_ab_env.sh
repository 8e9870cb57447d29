#!/bin/bash
# A/B of envelope-kernel objects on one box: scripts/gpu_ab_env.sh "objA objB objB ..." (run in that order)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() {
  scripts/env_variant.sh $1 -- python bench.py --workload config1 $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); m = d['stages']['device_ms']; print('$1', d['value'], d['ms_per_query'], 'env', m['envelopes'], 'stage1', m['stage1'], 'stage2', m['stage2'], 'host', m['host_domaindef'])"
}
for o in $1; do one $o; done
