#!/bin/bash
# What every stage of the cascade costs the line's workload: the same 4,000-profile slice with the filters closed one after the
# other (F1 = 1e-12: nothing passes MSV -> the MSV stage alone in the pipeline; F2 = 1e-12: MSV + bias + Viterbi; F3 = 1e-12: + the
# Forward parser; defaults: + Backward, regions, envelopes, ensembles, hit lists).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "--F1 1e-12" "--F2 1e-12" "--F3 1e-12" ""; do
  for i in ${RUNS:-1 2}; do
    python bench.py --gpus 1 --workload pfam --pfam-profiles ${PROFILES:-4000} --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline $v 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']
        print('[$v]', 'run $i', 'GCUPS', j['value'], 's', p['seconds'], 'stage counts', p['stage_counts_rank0'])
"
  done
done
