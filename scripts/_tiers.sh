set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s3
cd $R
timeout 500 python -m pytest tests/test_gpu_config3.py tests/test_gpu_search.py -x -q > gpurun_out/s3/tests.log 2>&1
tail -3 gpurun_out/s3/tests.log
timeout 300 python scripts/scan_ab.py msv_tiers 1 0 5 > gpurun_out/s3/scan_ab.log 2>&1
tail -4 gpurun_out/s3/scan_ab.log
timeout 300 python bench.py --workload pfam --no-cpu-baseline > gpurun_out/s3/pfam.json 2> gpurun_out/s3/pfam.err
python -c "
import json; d=json.loads(open('gpurun_out/s3/pfam.json').read().strip().splitlines()[-1]); print(d['value'], d.get('pfam',{}).get('seconds'))"
