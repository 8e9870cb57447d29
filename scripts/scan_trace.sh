#!/bin/bash
# kernel trace of a resident pass of the scan orientation (configs[2]):  scripts/scan_trace.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/${1:-gpurun_out/scan_trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $O/kt -o t -- python $R/scripts/config3_scan.py 20000 trace > $O/scan.log 2>&1
cd $R
python scripts/rocprof_dump_kernels.py $(find $O/kt -name "*.db" | head -1) $O/kernels.csv 0.7 > /dev/null
rm -rf $O/kt
python scripts/scan_trace_summary.py $O/kernels.csv $O/scan_trace.md "scan orientation (configs[2]: 20,000 profiles x 2,100 proteins), kernel trace of a resident pass: scripts/config3_scan.py 20000 trace under rocprofv3 --kernel-trace" | head -30
grep "hmmscan, \|traced pass" $O/scan.log
