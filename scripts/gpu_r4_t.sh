#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_config3.py -q -x 2>&1 | tail -3
for i in 1 2 3; do
  timeout 900 python bench.py --workload pfam --no-cpu-baseline --pfam-profiles 10000 --steps 2 --warmup 1 --spinup-max 1 > gpurun_out/r4t/pfam.$i.txt 2>/dev/null
done
for i in 1 2; do
  timeout 900 python bench.py --workload scan --no-cpu-baseline --steps 15 --warmup 3 > gpurun_out/r4t/scan.$i.txt 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4t/*.txt")):
    for line in open(f):
        if line.startswith('{'):
            j=json.loads(line); p=j.get('pfam'); s=j.get('scan')
            if p: print(f.split('/')[-1], "pfam seconds", p['seconds'], "GCUPS", p['value'], "msv_kernel", p['batch_ms_mean_rank0']['msv_kernel'], "stage1", p['batch_ms_mean_rank0']['stage1'])
            if s: print(f.split('/')[-1], "headline", j['value'], "scan", s['value'], s['seconds'])
PY
