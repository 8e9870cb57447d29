#!/bin/bash
out=gpurun_out/batch_min.txt; : > $out
run() { echo "## min $1 targets $2" >> $out
  P7X_BATCH_MIN=$1 python bench.py --gpus 1 --workload pfam --pfam-profiles 8000 --steps 8 --warmup 1 --no-cpu-baseline --pfam-targets $2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']; b = p['batch_ms_mean_rank0']
        print('GCUPS', j['value'], 's', p['seconds'], 'batchq', round(b['batch_queries'],1), 'stage1', round(b['stage1'],2), 'stage2', round(b['stage2'],2))
" >> $out; }
run 0 500000
run 64 500000
run 96 500000
run 0 500000
run 64 500000
run 64 250000
run 0 250000
run 64 125000
run 0 125000
cat $out
