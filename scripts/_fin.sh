set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s7
cd $R
timeout 300 python scripts/config3_scan.py 20000 trace > gpurun_out/s7/scan.log 2>&1
grep -c . gpurun_out/s7/scan.log
