#!/bin/bash
# A/B of the near-tie guard on the headline workload, same box
mkdir -p gpurun_out/r3d
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3d/build.log 2>&1
for g in 0 3e-6 0 3e-6; do
  timeout 300 python bench.py --steps 20 --warmup 5 --workload config1 --no-cpu-baseline --oa-guard $g > gpurun_out/r3d/bench_$g.json 2>> gpurun_out/r3d/bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3d/bench_$g.json').read().strip().splitlines()[-1])
print('guard $g', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stages']['device_ms'].items()})
PY
done
