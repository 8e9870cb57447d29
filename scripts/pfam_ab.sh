#!/bin/bash
# A/B of library options on the line's workload (a 4,000-profile slice of it): scripts/pfam_ab.sh "<opt>=<v> ..." ...
# every argument is one variant (a space-separated list of --debug-option settings; "-" = defaults); two runs each.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for variant in "$@"; do
  opts=""; [ "$variant" != "-" ] && for o in $variant; do opts="$opts --debug-option $o"; done
  for i in 1 2; do
    python bench.py --gpus 1 --workload pfam --pfam-profiles 4000 --steps 4 --warmup 1 --no-cpu-baseline $opts 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']; b = p['batch_ms_mean_rank0']
        print('$variant', 'run $i', 'GCUPS', j['value'], 's', p['seconds'], 'msv', round(b['msv'],2), 'vit', round(b['viterbi'],2), 'fwd', round(b['forward'],2), 'stage1', round(b['stage1'],2), 'stage2', round(b['stage2'],2), 'host_busy', round(b['host_stage_busy'],2))
"
  done
done
