#!/bin/bash
# round 4, late: the cleaned-up SSV kernel and the packed Viterbi stride again (tests), and a kernel trace of the scan orientation
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
( timeout 420 python -m pytest tests/test_gpu_longtarget.py -x -q -k "ssv or nhmmer_bmyd" 2>&1 | tail -8 ) > $O/w_ssv_tests.log
( timeout 400 python -m pytest tests/test_gpu_filters.py -x -q -k "wavefront_kernel_instantiation or viterbi_forward" 2>&1 | tail -8 ) > $O/w_vit_tests.log
timeout 150 python scripts/nhmmer_bench.py 250 50 > $O/w_ssv_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/$O/w_scan -o t -- python $R/scripts/config3_scan.py 20000 > $R/$O/w_scan.log 2>&1
cd $R
ls -la $O/w_scan/* | head
tail -n 3 $O/w_ssv_tests.log $O/w_vit_tests.log | cat; grep -h "run 1" $O/w_ssv_bench.log | cut -c1-230; tail -n 4 $O/w_scan.log | cut -c1-300
