#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export P7X_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 8 --workload pfam --pfam-profiles 4000 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/reh8.json 2> gpurun_out/reh8.err
echo rc $?
grep -v "amdgpu.ids\|hostname of the client" gpurun_out/reh8.err | tail -30
cat gpurun_out/reh8.json | cut -c1-300
