#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4s
cp pyhmmer_amd/libp7x.so /tmp/libp7x_orig.so
show() { python - "$1" <<'PY'
import json,sys
name=sys.argv[1]
for line in open(f"gpurun_out/r4s/{name}.txt"):
    if line.startswith('{'):
        j=json.loads(line)
        print(name, "headline", j['value'], "scan", (j.get('scan') or {}).get('value'), (j.get('scan') or {}).get('seconds'))
PY
}
for i in 1 2; do
  for v in side3 side1; do
    timeout 900 bash scripts/pipe_variant.sh scratch_variants/pipe_$v.o -- python bench.py --workload scan --no-cpu-baseline --steps 15 --warmup 3 > gpurun_out/r4s/$v.$i.txt 2>/dev/null; show $v.$i
  done
  cp /tmp/libp7x_orig.so pyhmmer_amd/libp7x.so
  timeout 900 python bench.py --workload scan --no-cpu-baseline --steps 15 --warmup 3 > gpurun_out/r4s/side7.$i.txt 2>/dev/null; show side7.$i
done
