#!/bin/bash
out=gpurun_out/scan_vitcut.txt; : > $out
scan() { echo "## scan $*" >> $out; python bench.py --gpus 1 --workload scan --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); s = j['scan']
        print('4k block', s['value'], s['passes_seconds_rank0'], 'fixture', s['fixture_proteome']['value'], s['fixture_proteome']['passes_seconds_rank0'])
" >> $out; }
scan
scan --debug-option vit_long_cut=1500
scan --debug-option vit_long_cut=3000
scan --debug-option vit_long_cut=100000
cat $out
