#!/bin/bash
# bench headline under several settings (each "VAR=val VAR=val|flags"), each run $N times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${N:-3}
while IFS= read -r line; do
  [ -z "$line" ] && continue
  envs="${line%%|*}"; flags="${line#*|}"
  for i in $(seq $N); do
    env $envs python bench.py --workload config1 $flags 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); m = d['stages']['device_ms']; print('''$line''', d['value'], d['ms_per_query'], 'env', m['envelopes'], 'stage1', m['stage1'], 'stage2', m['stage2'], 'host', m['host_domaindef'])"
  done
done
