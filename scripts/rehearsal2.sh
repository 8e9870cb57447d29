#!/bin/bash
# two ranks on one device (P7X_BENCH_SHARE_DEVICE=1): the driver's N > 1 command shape, the whole default line (all fields), short
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export P7X_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 3 --warmup 1 --pfam-profiles 4000 > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err
echo rc $?
python - <<PY
import json
for l in open("gpurun_out/rehearsal2.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["n_gpus"], d["scaling"], d["pfam"]["merge_seconds_rank0"], "c1", d["config1"]["value"], "scan", d["scan"]["value"], "nh", d["nhmmer"]["s_per_search"])
PY
tail -3 gpurun_out/rehearsal2.err
