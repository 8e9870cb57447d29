#!/bin/bash
# what ONE rank of an N-GPU run of the line's workload sees: the same 4,000-profile slice against 1/N of the targets
# (strong scaling: targets sharded by residues), with the library's feeders and with more
out=gpurun_out/shard_scaling.txt; : > $out
run() { echo "## targets $1 feeders ${2:-default}" >> $out
  python bench.py --gpus 1 --workload pfam --pfam-profiles 4000 --steps 4 --warmup 1 --no-cpu-baseline --pfam-targets $1 ${2:+--feeders $2} ${3:+--pipeline-depth $3} 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); p = j['pfam']; b = p['batch_ms_mean_rank0']
        print('GCUPS', j['value'], 's', p['seconds'], 'batchq', round(b['batch_queries'],1), 'msv', round(b['msv'],2), 'vit', round(b['viterbi'],2), 'fwd', round(b['forward'],2), 'stage1', round(b['stage1'],2), 'stage2', round(b['stage2'],2))
" >> $out; }
run 500000
run 250000
run 125000
run 62500
run 62500 4
run 62500 6 12
run 125000 4
cat $out
