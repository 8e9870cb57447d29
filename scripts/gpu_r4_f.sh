#!/bin/bash
# finishers / pipeline depth under the new host stage: headline and the many-profile stream
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4f
for fd in "2 4" "4 4" "6 6" "8 8"; do
  set -- $fd
  timeout 600 python bench.py --workload config1 --steps 12 --warmup 3 --no-cpu-baseline --finishers $1 --pipeline-depth $2 > gpurun_out/r4f/c1_f$1_d$2.txt 2>/dev/null
  python - $1 $2 <<'PY'
import json,sys
f,d=sys.argv[1:3]
for line in open(f"gpurun_out/r4f/c1_f{f}_d{d}.txt"):
    if line.startswith('{'):
        j=json.loads(line); print("config1 finishers",f,"depth",d,"GCUPS",j['value'],"ms/step",j['ms_per_step'])
PY
done
for fd in "2 4" "4 4" "6 6"; do
  set -- $fd
  timeout 900 python bench.py --workload pfam --no-cpu-baseline --pfam-finishers $1 --pfam-depth $2 > gpurun_out/r4f/pfam_f$1_d$2.txt 2>/dev/null
  python - $1 $2 <<'PY'
import json,sys
f,d=sys.argv[1:3]
for line in open(f"gpurun_out/r4f/pfam_f{f}_d{d}.txt"):
    if line.startswith('{'):
        j=json.loads(line); p=j.get('pfam',{}); print("pfam finishers",f,"depth",d,{k:p.get(k) for k in ('value','seconds','hits','finishers')})
PY
done
