#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export P7X_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for f in "" "--feeders 2"; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 8 --workload pfam --pfam-profiles 4000 --steps 4 --warmup 1 --no-cpu-baseline $f 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['pfam']; print('feeders [$f]', d['value'], p['seconds'], p['search_seconds_rank0'], p['merge_seconds_rank0'], p['feeders'])
"
done
