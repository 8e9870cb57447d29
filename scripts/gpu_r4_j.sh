#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4j
cp pyhmmer_amd/libp7x.so /tmp/libp7x_orig.so
show() { python - "$1" <<'PY'
import json,sys
name=sys.argv[1]
for line in open(f"gpurun_out/r4j/{name}.txt"):
    if line.startswith('{'):
        d=json.loads(line); dm=d['stages']['device_ms']
        print(name, "GCUPS", d['value'], "ms/step", d['ms_per_step'], "hits", d['stages']['hits'], "msv_kernel", dm['msv_kernel'], "stage2", dm['stage2'], "devwait", d['ranks']['per_rank'][0]['feeder_device_wait_frac'])
PY
}
B="python bench.py --workload config1 --steps 15 --warmup 3 --no-cpu-baseline"
timeout 600 $B > gpurun_out/r4j/default.txt 2>/dev/null; show default
timeout 600 $B --host-ensembles > gpurun_out/r4j/hostens.txt 2>/dev/null; show hostens
timeout 600 $B > gpurun_out/r4j/default2.txt 2>/dev/null; show default2
for v in noprio lowstream both; do
  timeout 600 bash scripts/ens_variant.sh scratch_variants/ens_$v.o -- $B > gpurun_out/r4j/$v.txt 2>/dev/null; show $v
done
cp /tmp/libp7x_orig.so pyhmmer_amd/libp7x.so
timeout 600 $B --pipeline-depth 12 > gpurun_out/r4j/d12.txt 2>/dev/null; show d12
timeout 600 $B --feeders 3 > gpurun_out/r4j/f3.txt 2>/dev/null; show f3
