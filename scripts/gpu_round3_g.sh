#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03c
python bench.py --gpus 1 --steps 20 --warmup 5 --workload config1 > gpurun_out/r03c/bench_config1.json 2> gpurun_out/r03c/err1
python bench.py --gpus 1 --steps 20 --warmup 5 --workload scan > gpurun_out/r03c/bench_scan.json 2> gpurun_out/r03c/err2
python - <<PY
import json
for f in ("bench_config1", "bench_scan"):
    d = json.loads([l for l in open(f"gpurun_out/r03c/{f}.json") if l.startswith("{")][0])
    print(f, d["value"], d["ms_per_query"], d["ms_per_step"], d.get("scan", {}).get("value"), d.get("scan", {}).get("seconds"))
PY
