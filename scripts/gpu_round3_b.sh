#!/bin/bash
mkdir -p gpurun_out/r3b
timeout 600 python scripts/oa_guard_sweep.py > gpurun_out/r3b/sweep.log 2>&1; echo "sweep rc $?" >> gpurun_out/r3b/rc.log
tail -n 30 gpurun_out/r3b/sweep.log
