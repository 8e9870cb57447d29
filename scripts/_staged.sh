set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s4
cd $R
timeout 500 python -m pytest tests/test_gpu_config3.py tests/test_gpu_search.py -x -q > gpurun_out/s4/tests.log 2>&1
tail -3 gpurun_out/s4/tests.log
timeout 400 python scripts/scan_ab.py stage_merge 1 0 5 > gpurun_out/s4/scan_ab.log 2>&1
tail -4 gpurun_out/s4/scan_ab.log
