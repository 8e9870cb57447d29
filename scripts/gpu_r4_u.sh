#!/bin/bash
# scan inside the full bench 0.53-0.59 s, alone 0.40-0.41 s: is it what ran before it (load history), or the process state?
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4u
timeout 900 python bench.py --workload scan --no-cpu-baseline --steps 15 --warmup 3 > gpurun_out/r4u/scan_after_3s.txt 2>/dev/null
timeout 900 python bench.py --workload scan --no-cpu-baseline --steps 200 --warmup 3 > gpurun_out/r4u/scan_after_30s_headline.txt 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4u/*.txt")):
    for line in open(f):
        if line.startswith('{'):
            j=json.loads(line); s=j['scan']
            print(f.split('/')[-1], "headline", j['value'], "scan", s['value'], s['seconds'], {k:v for k,v in s.items() if k in ('batch_ms_mean_rank0','setup_seconds')})
PY
rocm-smi --showclocks --showpower 2>/dev/null | head -20
