#!/bin/bash
# The measurements behind profiles/rNN_*: the driver's bench command, the same under rocprofv3 --kernel-trace --stats, and
# the PMC passes of the headline workload at the batch size the bench uses.   usage: scripts/round_profiles.sh <outdir> <B>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-7}
mkdir -p $R/$OUT
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$OUT/bench_traced.json 2> $R/$OUT/bench_traced.err
cd $R
python scripts/rocprof_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/kernel_stats.md "python bench.py --gpus 1 --steps 20 --warmup 5 under rocprofv3 --kernel-trace --stats" > /dev/null
scripts/pmc_msv.sh $OUT/pmc $B > /dev/null
find $OUT -name "*.db" -size +20M -delete
ls -la $OUT
