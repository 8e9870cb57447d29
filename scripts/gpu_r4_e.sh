#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4e
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r4e/gputests.txt
tail -4 gpurun_out/r4e/gputests.txt
bash scripts/gpu_r4_d.sh
timeout 600 python bench.py --debug-option trace_finish=1 --workload config1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4e/bench_dbg.txt 2> gpurun_out/r4e/bench_dbg.err
grep "finish\] nq 7" gpurun_out/r4e/bench_dbg.err | tail -6
