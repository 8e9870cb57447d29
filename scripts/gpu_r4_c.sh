#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_ensembles.py -q -x 2>&1 | tail -25 > gpurun_out/r4c/ens.txt
tail -3 gpurun_out/r4c/ens.txt
timeout 600 python bench.py --debug-option trace_finish=1 --workload config1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4c/bench_config1.txt 2> gpurun_out/r4c/bench_config1.err
tail -c 300 gpurun_out/r4c/bench_config1.txt
grep "finish\]" gpurun_out/r4c/bench_config1.err | tail -4
timeout 600 bash scripts/ens_variant.sh scratch_variants/ens_prof.o -- python scripts/ens_phase_profile.py > gpurun_out/r4c/phases.txt 2>&1
tail -9 gpurun_out/r4c/phases.txt
