#!/bin/bash
# PMC passes over the headline workload with nothing else in flight (scripts/config1_phases.py), one counter group per run.
# usage (on the GPU box): scripts/pmc_msv.sh <outdir> [B]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-4}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $R/$OUT/p$i -o pmc -- python $R/scripts/config1_phases.py $B 2 > $R/$OUT/p$i.log 2>&1
done
cd $R
python scripts/rocprof_pmc_summary.py $OUT/summary.md "config1_phases.py $B 2 under rocprofv3 --pmc" $(find $OUT -name "*.db") > /dev/null
grep -A30 "msv_fast" $OUT/summary.md | head -45
