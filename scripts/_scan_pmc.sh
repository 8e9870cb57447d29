set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s2
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $R/gpurun_out/s2/pmc -o pmc -- python $R/scripts/config3_scan.py 20000 > $R/gpurun_out/s2/scan_pmc.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob, re, collections
db=glob.glob("gpurun_out/s2/pmc/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(counters_collection)")]
kcol="kernel_name" if "kernel_name" in cols else "name"
ccol="counter_name" if "counter_name" in cols else "pmc_name"
vcol="value" if "value" in cols else "counter_value"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.Counter()
for k,cn,v,n in c.execute(f"select {kcol},{ccol},sum({vcol}),count(distinct dispatch_id) from counters_collection group by {kcol},{ccol}"):
    f=re.sub(r"<.*","",k.replace("void p7x::","").replace("p7x::","")).split("(")[0]
    agg[f][cn]+=v; nd[(f,cn)]+=n
with open("gpurun_out/s2/pmc_summary.txt","w") as o:
    for f,d in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_ACTIVE_INST_VALU",0)):
        o.write(f"{f:34s} n={nd[(f,'SQ_WAVES')]:6d} " + " ".join(f"{cn}={v:.4g}" for cn,v in sorted(d.items())) + "\n")
print(open("gpurun_out/s2/pmc_summary.txt").read())
PY
rm -rf gpurun_out/s2/pmc
