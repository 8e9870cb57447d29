#!/bin/bash
# Per-kernel times of the line's workload (configs[3], the Pfam-shaped library x 500,000 targets) under rocprofv3
# --kernel-trace --stats: scripts/pfam_profile.sh <outdir under gpurun_out> [profiles]   (the first <profiles> library entries,
# default 2000: an untimed pass over them, then two steps of 1,000; durations are stretched by the batches in flight)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/pfam_prof}; NP=${2:-2000}
O=$R/$OUT; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o pfam -- python $R/bench.py --gpus 1 --workload pfam --pfam-profiles $NP --steps 2 --warmup 0 --no-cpu-baseline > $O/pfam_traced.json 2> $O/pfam_traced.err
cd $R
python scripts/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/pfam_kernel_stats.md "python bench.py --gpus 1 --workload pfam --pfam-profiles $NP --steps 2 --warmup 0 --no-cpu-baseline under rocprofv3 --kernel-trace --stats (the calibration of the library's profiles, an untimed pass and two timed steps; durations stretched by the batches in flight)" > /dev/null
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
head -45 $O/pfam_kernel_stats.md | cut -c1-200
tail -c 400 $O/pfam_traced.err
