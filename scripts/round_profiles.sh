#!/bin/bash
# The measurements behind profiles/rNN_*, on the GPU box:  scripts/round_profiles.sh <outdir under gpurun_out> [B]
#   1. the GPU tests and smoke();  2. the driver's bench command (twice);  3. the headline workload under rocprofv3
#   --kernel-trace --stats (the full command dies inside rocprofv3's interception layer once a few dozen streams are live);
#   4. one batch of B queries at a time with nothing else in flight under --kernel-trace: stand-alone kernel durations;
#   5. PMC passes of that (one counter group per run; FETCH_SIZE / WRITE_SIZE in runs of their own).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/round}; B=${2:-7}
O=$R/$OUT; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_run$i.json 2> $O/bench_run$i.err; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload config1 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_alone -o alone -- python $R/scripts/config1_phases.py $B 4 > $O/alone.log 2>&1
cd $R
python scripts/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/kernel_stats.md "python bench.py --gpus 1 --steps 20 --warmup 5 --workload config1 --no-cpu-baseline under rocprofv3 --kernel-trace --stats (durations stretched by the overlap of eight batches in flight)" > /dev/null
python scripts/rocprof_summary.py $(find $O/trace_alone -name "*.db" | head -1) $O/kernel_stats_alone.md "python scripts/config1_phases.py $B 4 under rocprofv3 --kernel-trace --stats: one batch of $B queries at a time, nothing else in flight (stand-alone kernel durations)" > /dev/null
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); timeout 200 rocprofv3 --pmc $grp -d $O/pmc/p$i -o pmc -- python $R/scripts/config1_phases.py $B 2 > $O/pmc_p$i.log 2>&1
done
cd $R
python scripts/rocprof_pmc_summary.py $O/pmc_summary.md "config1_phases.py $B 2 under rocprofv3 --pmc (one counter group per run)" $(find $O/pmc -name "*.db") > /dev/null
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
cat $O/tests.log $O/smoke.log | cut -c1-300
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_run*.json")):
    for line in open(f):
        if line.startswith('{'):
            j=json.loads(line)
            print(f.split('/')[-1], "headline", j['value'], j['ms_per_step'], "pfam", (j.get('pfam') or {}).get('value'), (j.get('pfam') or {}).get('seconds'), "scan", (j.get('scan') or {}).get('value'), (j.get('scan') or {}).get('passes_seconds_rank0'), "nhmmer", (j.get('nhmmer') or {}).get('s_per_search'), (j.get('nhmmer') or {}).get('s_one_search_alone'), "cpu", (j.get('cpu_baseline') or {}).get('value'))
PY
head -14 $O/kernel_stats_alone.md | cut -c1-170
