#!/bin/bash
# config1 (one profile x 1M targets) with the near-tie guards switched off one after the other: what each costs the stream
out=gpurun_out/c1_guard_ab.txt; : > $out
run() { echo "## $*" >> $out; python bench.py --workload config1 --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); s = d['stages']['device_ms']
print(d['value'], d['ms_per_step'], d['config']['spinup_windows_s'], {k: s[k] for k in ('stage1','stage2','envelopes','host_stage_busy','ensemble_wait','envelope_wait','host_multi')}, d['ranks']['per_rank'][0]['feeder_device_wait_frac'])" >> $out; }
run
run --oa-guard 0
run --ens-guard 0
run --debug-option region_guard_ppm=0
run --oa-guard 0 --ens-guard 0 --debug-option region_guard_ppm=0
run
cat $out
