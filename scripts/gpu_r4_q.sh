#!/bin/bash
# hardware queues: does the run-to-run bimodality of the many-profile stream come from streams sharing HSA queues?
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4q
for q in 8 16 24 8 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --workload pfam --no-cpu-baseline --pfam-profiles 10000 --steps 2 --warmup 1 --spinup-max 1 > gpurun_out/r4q/pfam_q$q.$RANDOM.txt 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4q/pfam_q*.txt")):
    for line in open(f):
        if line.startswith('{'):
            j=json.loads(line); p=j['pfam']; b=p['batch_ms_mean_rank0']
            print(f.split('/')[-1], "seconds", p['seconds'], "GCUPS", p['value'], "msv_kernel", b['msv_kernel'], "viterbi", b['viterbi'], "stage1", b['stage1'], "stage2", b['stage2'])
PY
