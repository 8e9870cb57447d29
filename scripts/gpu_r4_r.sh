#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4r
cp pyhmmer_amd/libp7x.so /tmp/libp7x_orig.so
B="python bench.py --workload pfam --no-cpu-baseline --pfam-profiles 10000 --steps 2 --warmup 1 --spinup-max 1"
for i in 1 2 3; do
  for v in side3 side1; do
    timeout 900 bash scripts/pipe_variant.sh scratch_variants/pipe_$v.o -- $B > gpurun_out/r4r/pfam_$v.$i.txt 2>/dev/null
  done
  cp /tmp/libp7x_orig.so pyhmmer_amd/libp7x.so
  timeout 900 $B > gpurun_out/r4r/pfam_side7.$i.txt 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4r/pfam_*.txt")):
    for line in open(f):
        if line.startswith('{'):
            j=json.loads(line); p=j['pfam']; b=p['batch_ms_mean_rank0']
            print(f.split('/')[-1], "seconds", p['seconds'], "GCUPS", p['value'], "msv_kernel", b['msv_kernel'], "viterbi", b['viterbi'], "stage1", b['stage1'], "stage2", b['stage2'], "headline", j['value'])
PY
