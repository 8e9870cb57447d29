#!/bin/bash
# A/B of the headline: ensembles on the device vs on the host workers, with 16 and with 8 host threads
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4d
run() { # name, extra args...
  name=$1; shift
  timeout 600 "$@" > gpurun_out/r4d/$name.txt 2> gpurun_out/r4d/$name.err
  python - "$name" <<'PY'
import json,sys
name=sys.argv[1]
for line in open(f"gpurun_out/r4d/{name}.txt"):
    if line.startswith('{'):
        d=json.loads(line); dm=d['stages']['device_ms']
        print(name, "GCUPS", d['value'], "ms/step", d['ms_per_step'], "hits", d['stages']['hits'], "host_domaindef", dm['host_domaindef'], "host_multi", dm['host_multi'], "envelopes", dm['envelopes'], "stage1", dm['stage1'], "stage2", dm['stage2'])
PY
}
B="python bench.py --workload config1 --steps 12 --warmup 3 --no-cpu-baseline"
run dev16 $B
run host16 $B --host-ensembles
run dev8 taskset -c 0-7 $B
run host8 taskset -c 0-7 $B --host-ensembles
run dev16_f2 $B --finishers 2
run dev16_d6 $B --finishers 6 --pipeline-depth 6
