#!/bin/bash
# same box, back to back: round 3's tree (its own bench and defaults) and this round's
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out/r4k
show() { python - "$1" <<'PY'
import json,sys
name=sys.argv[1]
for line in open(f"/root/repo/gpurun_out/r4k/{name}.txt"):
    if line.startswith('{'):
        d=json.loads(line); dm=d['stages']['device_ms']
        print(name, "GCUPS", d['value'], "ms/step", d['ms_per_step'], "hits", d['stages']['hits'], {k: round(dm[k],2) for k in ('msv_kernel','viterbi','forward','fwd_rows','bias','envelopes','host_multi','stage1','stage2')})
PY
}
for i in 1 2; do
cd $R/scratch_variants/r03 && timeout 600 python bench.py --workload config1 --steps 15 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r4k/r03_$i.txt 2>/dev/null; show r03_$i
cd $R && timeout 600 python bench.py --workload config1 --steps 15 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r4k/r04_$i.txt 2>/dev/null; show r04_$i
done
