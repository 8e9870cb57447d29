#!/bin/bash
out=gpurun_out/c1_feeders.txt; : > $out
run() { echo "## $*" >> $out; python bench.py --workload config1 --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); s = d['stages']['device_ms']
print(d['value'], d['ms_per_step'], d['config']['spinup_windows_s'][-3:], {k: s[k] for k in ('stage1','stage2')}, d['ranks']['per_rank'][0]['feeder_device_wait_frac'])" >> $out; }
run --feeders 2
run --feeders 3
run --feeders 2
run --feeders 3
cat $out
