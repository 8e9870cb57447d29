#!/bin/bash
out=gpurun_out/batch_cells.txt; : > $out
run() { echo "## cells $1 ${2}" >> $out
  P7X_BATCH_CELLS=$1 python bench.py --gpus 1 --workload pfam --pfam-profiles 8000 --steps 8 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']; b = p['batch_ms_mean_rank0']
        print('GCUPS', j['value'], 's', p['seconds'], 'feeders', p['feeders'], 'batchq', round(b['batch_queries'],1), 'stage1', round(b['stage1'],2), 'stage2', round(b['stage2'],2))
" >> $out; }
run 3e11
run 6e11
run 1.2e12
run 2.4e12
run 6e11 "--pfam-finishers 12 --pfam-depth 12"
cat $out
