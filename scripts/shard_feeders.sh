#!/bin/bash
# feeders by shard size: 8,000 profiles of the line's library against 1/2, 1/4, 1/8 of its targets
out=gpurun_out/shard_feeders.txt; : > $out
run() { echo "## targets $1 feeders $2" >> $out
  python bench.py --gpus 1 --workload pfam --pfam-profiles 8000 --steps 8 --warmup 1 --no-cpu-baseline --pfam-targets $1 --feeders $2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']; b = p['batch_ms_mean_rank0']
        print('GCUPS', j['value'], 's', p['seconds'], 'batchq', round(b['batch_queries'],1), 'stage1', round(b['stage1'],2), 'stage2', round(b['stage2'],2))
" >> $out; }
for t in 250000 125000 62500; do for f in 2 3 4; do run $t $f; done; done
cat $out
