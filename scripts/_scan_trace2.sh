set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s5
cd $R
timeout 300 python scripts/scan_sweep.py 20000 0,3,4,1 0,6,8,1 0,4,8,2 2048,6,10,1 1024,8,12,1 > gpurun_out/s5/sweep.log 2>&1
tail -5 gpurun_out/s5/sweep.log
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $R/gpurun_out/s5/kt -o t -- python $R/scripts/config3_scan.py 20000 trace > $R/gpurun_out/s5/scan.log 2>&1
cd $R
DB=$(ls gpurun_out/s5/kt/*.db gpurun_out/s5/kt/*/*.db 2>/dev/null | head -1)
python scripts/rocprof_dump_kernels.py $DB gpurun_out/s5/kernels.csv 0.6
rm -rf gpurun_out/s5/kt
