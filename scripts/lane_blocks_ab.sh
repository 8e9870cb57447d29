#!/bin/bash
# blocks per lane of the MSV launches (lane_grid_pull): the scan orientation and the line's workload with the old and the new rule
out=gpurun_out/lane_blocks_ab.txt; : > $out
scan() { echo "## scan $*" >> $out; python bench.py --gpus 1 --workload scan --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); s = j['scan']
        print('4k block', s['value'], s['passes_seconds_rank0'], 'fixture', s['fixture_proteome']['value'], s['fixture_proteome']['passes_seconds_rank0'])
" >> $out; }
scan --debug-option msv_lane_blocks=0
scan
scan --debug-option msv_lane_blocks=0
scan
echo "## pfam slice" >> $out
bash scripts/pfam_ab.sh "msv_lane_blocks=0" "-" >> $out 2>&1
cat $out
