#!/bin/bash
# round 4, late: the quad SSV kernel (tests, three kernels timed), the packed Viterbi kernel's LDS image variants, scan trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
( timeout 420 python -m pytest tests/test_gpu_longtarget.py -x -q -k "ssv" 2>&1 | tail -15 ) > $O/v_ssv_tests.log
for v in 1 3 4; do timeout 150 python scripts/nhmmer_bench.py 250 50 $v > $O/v_ssv_bench_$v.log 2>&1; done
( timeout 400 python -m pytest tests/test_gpu_filters.py -x -q -k "wavefront_kernel_instantiation or viterbi_forward" 2>&1 | tail -8 ) > $O/v_vit_tests_c2.log
timeout 150 python scripts/config1_phases.py 7 4 > $O/v_phases_c2.log 2>&1
for v in c0 c1 c2w8; do
  scripts/obj_variant.sh p7x_vitpk.hip scratch_variants/vitpk_$v.o -- timeout 150 python scripts/config1_phases.py 7 4 > $O/v_phases_$v.log 2>&1
done
scripts/obj_variant.sh p7x_vitpk.hip - -- true
timeout 200 python scripts/config3_scan.py 20000 trace > $O/v_scan_trace.log 2>&1
tail -3 $O/v_ssv_tests.log $O/v_vit_tests_c2.log; grep -h "run 1" $O/v_ssv_bench_*.log | cut -c1-230; grep -h "batch of" $O/v_phases_*.log | tail -n 8 | cut -c1-400
