#!/bin/bash
# round 4, late: two of the eight queue positions kept for the ensemble streams, against the same library without (ensq0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
run() {  # label
  timeout 200 python scripts/config3_scan.py 20000 trace 2> $O/y_scan_$1.err | grep -v "^\[" > $O/y_scan_$1.log
  timeout 200 python bench.py --workload config1 --no-cpu-baseline --steps 10 --warmup 3 > $O/y_head_$1.log 2>&1
}
run new
timeout 300 python bench.py --workload pfam --no-cpu-baseline > $O/y_pfam_new.log 2>&1
scripts/obj_variant.sh p7x_device.hip scratch_variants/device_ensq0.o -- true; run ensq0
scripts/obj_variant.sh p7x_device.hip - -- true
for v in new ensq0; do echo "== $v"; grep "hmmscan\|traced" $O/y_scan_$v.log | cut -c1-120; grep "^\[finish\]" $O/y_scan_$v.err | sed -E 's/.*(ens_wait [0-9.]+).*/\1/' | tr '\n' ' '; echo; python - <<PY
import json
for l in open("$O/y_head_$v.log"):
    if l.startswith("{"):
        d=json.loads(l); print("headline", d["value"], d["ms_per_step"], d.get("batch_ms_mean_rank0", d.get("config",{})) if False else "")
PY
done
python - <<PY
import json
for l in open("$O/y_pfam_new.log"):
    if l.startswith("{"):
        d=json.loads(l); p=d.get("pfam",{}); print("pfam", p.get("value"), p.get("seconds"), {k:v for k,v in p.get("batch_ms_mean_rank0",{}).items() if k in ("ensemble_wait","envelope_wait","host_stage_busy","stage1","stage2")})
PY
