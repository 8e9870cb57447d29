#!/bin/bash
# round 4, final tree: the GPU tests and smoke(), the driver's bench command three times, the headline under rocprofv3
# --kernel-trace --stats, PMC passes of the headline batch (three groups) and of the nhmmer scan
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r04f; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_run$i.json 2> $O/bench_run$i.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04f/bench_run*.json")):
    for line in open(f):
        if line.startswith('{'):
            j=json.loads(line)
            print(f.split('/')[-1], "headline", j['value'], "ms/step", j['ms_per_step'], "pfam", (j.get('pfam') or {}).get('value'), (j.get('pfam') or {}).get('seconds'),
                  "scan", (j.get('scan') or {}).get('value'), (j.get('scan') or {}).get('seconds'), "nhmmer", (j.get('nhmmer') or {}).get('s_per_search'), "cpu", (j.get('cpu_baseline') or {}).get('value'), "roof", j.get('roofline'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload config1 --no-cpu-baseline > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err
cd $R
python scripts/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/kernel_stats.md "python bench.py --gpus 1 --steps 20 --warmup 5 --workload config1 --no-cpu-baseline under rocprofv3 --kernel-trace --stats" > /dev/null
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); timeout 200 rocprofv3 --pmc $grp -d $R/$O/pmc/p$i -o pmc -- python $R/scripts/config1_phases.py 7 2 > $R/$O/pmc_p$i.log 2>&1
done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1)); timeout 200 rocprofv3 --pmc $grp -d $R/$O/pmc_ssv/p$i -o pmc -- python $R/scripts/nhmmer_bench.py 100 > $R/$O/pmc_ssv_p$i.log 2>&1
done
cd $R
python scripts/rocprof_pmc_summary.py $O/pmc_summary.md "config1_phases.py 7 2 under rocprofv3 --pmc" $(find $O/pmc -name "*.db") > /dev/null
python scripts/rocprof_pmc_summary.py $O/pmc_ssv_summary.md "nhmmer_bench.py 100 under rocprofv3 --pmc" $(find $O/pmc_ssv -name "*.db") > /dev/null
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
cat $O/tests.log $O/smoke.log; head -12 $O/kernel_stats.md | cut -c1-150; grep -A12 "ssvlong" $O/pmc_ssv_summary.md | head -16; du -sh $O
