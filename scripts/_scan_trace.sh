set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s1
cd $R
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $R/gpurun_out/s1/kt -o t -- python $R/scripts/config3_scan.py 20000 trace > $R/gpurun_out/s1/scan.log 2>&1
cd $R
DB=$(ls gpurun_out/s1/kt/*.db gpurun_out/s1/kt/*/*.db 2>/dev/null | head -1)
python scripts/rocprof_dump_kernels.py $DB gpurun_out/s1/kernels.csv 0.6
rm -rf gpurun_out/s1/kt
tail -3 gpurun_out/s1/bench.json | cut -c1-600
