#!/bin/bash
# round 4, first GPU call: the new ensemble kernels against the host twin, the envelope tests under the host twin's new
# order of operations, a short headline run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_ensembles.py -q -x 2>&1 | tail -25 > gpurun_out/r4a/ens.txt
timeout 900 python -m pytest tests/test_gpu_envelopes.py -q 2>&1 | tail -25 > gpurun_out/r4a/env.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4a/smoke.txt 2>&1
timeout 600 python bench.py --workload config1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4a/bench_config1.txt 2>&1
tail -3 gpurun_out/r4a/ens.txt gpurun_out/r4a/env.txt gpurun_out/r4a/smoke.txt; tail -c 1500 gpurun_out/r4a/bench_config1.txt
