#!/bin/bash
# Multi-GPU rehearsal on ONE MI355X (no 8-GPU node is the builder's to use): the host side of both N-device paths.
#   1. bench.py --gpus 8 under torch.distributed with P7X_BENCH_SHARE_DEVICE=1: eight ranks (one process each, gloo for the
#      few scalars) share the device; the line's workload strong-scaled (every rank 1/8 of the 500,000 targets, a
#      4,000-profile slice), gather + merge_many on rank 0 timed.
#   2. bench.py --workload config1 --inproc-devices 8: ONE process, hmmer.hmmsearch over eight resident shards (the API's
#      default path on an eight-GPU node), feeder / finisher wait fractions.
# usage: scripts/rehearsal8.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/${1:-gpurun_out/rehearsal8}; mkdir -p $O; cd $R
export P7X_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 8 --workload pfam --pfam-profiles 4000 --steps 4 --warmup 1 --no-cpu-baseline > $O/gpus8_pfam.json 2> $O/gpus8_pfam.err
timeout 600 python bench.py --gpus 1 --workload pfam --pfam-profiles 4000 --steps 4 --warmup 1 --no-cpu-baseline > $O/gpus1_pfam.json 2> $O/gpus1_pfam.err
timeout 600 python bench.py --gpus 1 --workload config1 --nseq 125000 --inproc-devices 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/inproc8_config1.json 2> $O/inproc8_config1.err
timeout 600 python bench.py --gpus 1 --workload config1 --nseq 1000000 --steps 10 --warmup 2 --no-cpu-baseline > $O/inproc1_config1.json 2> $O/inproc1_config1.err
python - <<PY
import json
def line(f):
    for l in open(f):
        if l.startswith('{'): return json.loads(l)
    return None
o="$O"
with open(o + "/summary.md", "w") as w:
    w.write("# Multi-GPU rehearsal on one MI355X (P7X_BENCH_SHARE_DEVICE=1): eight shards, one device\n\n")
    a, b = line(o + "/gpus8_pfam.json"), line(o + "/gpus1_pfam.json")
    if a and b:
        pa, pb = a["pfam"], b["pfam"]
        w.write("## process per GPU: bench.py --gpus 8 --workload pfam (4,000 profiles x 500,000 targets, 8 ranks on one device) vs --gpus 1\n\n")
        w.write(f"* 8 ranks: {a['value']} GCUPS, {pa['seconds']} s; rank 0: search {pa['search_seconds_rank0']} s, merge {pa['merge_seconds_rank0']}\n")
        w.write(f"* 1 rank:  {b['value']} GCUPS, {pb['seconds']} s\n")
        w.write(f"* eight processes sharing the device reach {a['value'] / b['value']:.3f} of one process's throughput: what the sharding, eight host pipelines and the gather + merge cost when the device time is the same\n\n")
    c, d = line(o + "/inproc8_config1.json"), line(o + "/inproc1_config1.json")
    if c and d:
        w.write("## one process, eight shards: bench.py --workload config1 --inproc-devices 8 --nseq 125000 (8 x 125,000 targets) vs one block of 1,000,000\n\n")
        for name, j in (("8 shards", c), ("1 block", d)):
            r = j["ranks"]["per_rank"][0]
            w.write(f"* {name}: {j['value']} GCUPS, {j['ms_per_query']} ms per query; feeders: device wait {r['feeder_device_wait_frac']}, slot wait {r['feeder_slot_wait_frac']}, enqueue {r['feeder_enqueue_frac']}; host stage {r['host_stage_s_per_batch']} s per batch, {r['host_stages_in_flight']} in flight; config {dict((k, j['config'][k]) for k in ('pipeline_depth', 'feeders', 'finishers'))}\n")
print(open(o + "/summary.md").read())
PY
