#!/bin/bash
python -m pytest tests/test_gpu_envelopes.py tests/test_gpu_oracle_domains.py tests/test_gpu_search.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/env_tests.log
cat gpurun_out/env_tests.log
out=gpurun_out/c1_guard_ab2.txt; : > $out
run() { echo "## $*" >> $out; python bench.py --workload config1 --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); s = d['stages']['device_ms']
print(d['value'], d['ms_per_step'], d['config']['spinup_windows_s'], {k: s[k] for k in ('stage1','stage2','envelopes','host_stage_busy','ensemble_wait','envelope_wait','host_multi')}, d['ranks']['per_rank'][0]['feeder_device_wait_frac'], d['config']['latency_ms_one_query_idle_device'])" >> $out; }
run
run --oa-guard 0
run
cat $out
