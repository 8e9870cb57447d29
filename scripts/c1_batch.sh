#!/bin/bash
out=gpurun_out/c1_batch.txt; : > $out
run() { echo "## $*" >> $out; python bench.py --workload config1 --steps 12 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); s = d['stages']['device_ms']
print(d['value'], d['ms_per_step'], {k: round(s[k],1) for k in ('stage1','stage2','host_stage_busy')}, d['ranks']['per_rank'][0]['feeder_device_wait_frac'])" >> $out; }
run --batch 4
run --batch 7
run --batch 10
run --batch 14
run --batch 7 --pipeline-depth 12 --finishers 12
cat $out
