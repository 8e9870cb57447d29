#!/bin/bash
# kernel trace of the scan orientation with the reserved ensemble queues: which hardware queue every stream lands on
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/$O/z_scan -o t -- python $R/scripts/config3_scan.py 20000 trace > $R/$O/z_scan.log 2> $R/$O/z_scan.err
cd $R
grep "hmmscan\|traced" $O/z_scan.log | cut -c1-120; grep "^\[finish\]" $O/z_scan.err | sed -E 's/.*(ens_wait [0-9.]+).*/\1/' | tr '\n' ' '
