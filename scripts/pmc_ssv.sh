#!/bin/bash
# PMC passes over the nhmmer SSV scan (scripts/nhmmer_bench.py), one counter group per run.
# usage (on the GPU box): scripts/pmc_ssv.sh <outdir> [Mbp]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; MBP=${2:-100}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $R/$OUT/p$i -o pmc -- python $R/scripts/nhmmer_bench.py $MBP > $R/$OUT/p$i.log 2>&1
done
cd $R
python scripts/rocprof_pmc_summary.py $OUT/summary.md "nhmmer_bench.py $MBP under rocprofv3 --pmc" $(find $OUT -name "*.db") > /dev/null
grep -A22 "ssvlong" $OUT/summary.md | head -40
