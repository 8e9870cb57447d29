cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_longtarget.py -x -q -s 2>&1 | tail -15
timeout 300 python bench.py --workload nhmmer --steps 3 --warmup 1 2>&1 | tail -3
