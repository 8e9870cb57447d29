#!/bin/bash
out=gpurun_out/shard_feeders2.txt; : > $out
run() { echo "## targets $1 feeders $2 depth ${3:-default}" >> $out
  python bench.py --gpus 1 --workload pfam --pfam-profiles 8000 --steps 8 --warmup 1 --no-cpu-baseline --pfam-targets $1 --feeders $2 ${3:+--pfam-depth $3} 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']; b = p['batch_ms_mean_rank0']
        print('GCUPS', j['value'], 's', p['seconds'], 'batchq', round(b['batch_queries'],1), 'stage1', round(b['stage1'],2), 'stage2', round(b['stage2'],2))
" >> $out; }
run 500000 2
run 500000 3
run 500000 3 12
run 500000 2
run 500000 3
cat $out
