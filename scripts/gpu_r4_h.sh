#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4h
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -60 > gpurun_out/r4h/gputests.txt
tail -5 gpurun_out/r4h/gputests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err
python - <<'PY'
import json
for line in open("gpurun_out/r4h/bench.json"):
    if line.startswith('{'):
        j=json.loads(line)
        print("headline", j['value'], j['ms_per_step'], j['config'].get('pipeline_depth'), j['config'].get('finishers'))
        print("ranks", j.get('ranks'))
        print("cpu", j.get('cpu_baseline'))
        for k in ('pfam','scan','nhmmer'):
            v=j.get(k) or {}
            print(k, {kk:v.get(kk) for kk in ('value','seconds','s_per_search','s_one_search_alone','hits','ssv_scan_gcups','ms')})
PY
tail -3 gpurun_out/r4h/bench.err
