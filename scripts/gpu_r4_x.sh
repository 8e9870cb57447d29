#!/bin/bash
# round 4, late: which priority pools the cascade / ensemble streams live in (the ensemble kernels queued behind MSV kernels in the scan)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
run() {  # label
  timeout 200 python scripts/config3_scan.py 20000 trace 2> /dev/null | grep -v "^\[" > $O/x_scan_$1.log
  timeout 200 python bench.py --workload config1 --no-cpu-baseline --steps 10 --warmup 3 > $O/x_head_$1.log 2>&1
}
run base
scripts/obj_variant.sh p7x_ensemble.hip scratch_variants/ens_normal.o -- true; run ensnormal
scripts/obj_variant.sh p7x_ensemble.hip - -- true
scripts/obj_variant.sh p7x_device.hip scratch_variants/device_wsnormal.o -- true; run wsnormal
scripts/obj_variant.sh p7x_device.hip - -- true
for v in base ensnormal wsnormal; do echo "== $v"; grep "hmmscan\|traced" $O/x_scan_$v.log | cut -c1-120; python - <<PY
import json
for l in open("$O/x_head_$v.log"):
    if l.startswith("{"):
        d=json.loads(l); print("headline", d["value"], d["ms_per_step"])
PY
done
