#!/bin/bash
# round 4, late: transitions in two LDS planes in the envelope kernel: tests, time, LDS counters
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_envelopes.py tests/test_gpu_ensembles.py -x -q 2>&1 | tail -6 ) > $O/env_tests.log
timeout 150 python scripts/config1_phases.py 7 4 > $O/env_phases.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/$O/env_pmc -o pmc -- python $R/scripts/config1_phases.py 7 2 > $R/$O/env_pmc.log 2>&1
cd $R
python scripts/rocprof_pmc_summary.py $O/env_pmc_summary.md "config1_phases.py 7 2 under rocprofv3 --pmc" $(find $O/env_pmc -name "*.db") > /dev/null
tail -3 $O/env_tests.log; grep "batch of" $O/env_phases.log | sed -E "s/.*'envelopes': ([0-9.]+).*'stage2': ([0-9.]+).*/envelopes \1 stage2 \2/"; grep -A8 "env_kernel<5" $O/env_pmc_summary.md | head -12
find $O/env_pmc -name "*.db" -delete
