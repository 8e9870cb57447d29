#!/bin/bash
# stand-alone kernel durations of configs[1]: one batch of 7 queries at a time (and of 1 query), nothing else in flight
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_alone; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 7 1; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/t$b -o t -- python $R/scripts/config1_phases.py $b 4 > $O/b$b.log 2>&1
  cd $R
  python scripts/rocprof_summary.py $(find $O/t$b -name "*.db" | head -1) $O/kernels_alone_b$b.md "python scripts/config1_phases.py $b 4 under rocprofv3 --kernel-trace --stats: one batch of $b queries at a time, nothing else in flight (stand-alone kernel durations)" > /dev/null
  cd /tmp
done
cd $R; find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
head -24 $O/kernels_alone_b7.md | cut -c1-190; head -16 $O/kernels_alone_b1.md | cut -c1-190
