"""One rank's view of an N-GPU run of the line's workload: NP library profiles against NT targets through hmmer.hmmsearch with the
library's defaults, and where the query pipeline's threads spent the time (hmmer.pipeline_stats).  usage: shard_stats.py [NP] [NT] [feeders]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench_workloads as bw
from pyhmmer_amd import plan7, hmmer
nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
ntgt = int(sys.argv[2]) if len(sys.argv) > 2 else 62500
feeders = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hmms, lib_lengths, templates = bw.make_library(20000, device=0, count=nprof)
bg = plan7.Background(hmms[0].alphabet)
oms = [plan7.OptimizedProfile(h, bg, 400) for h in hmms]
flat, offsets, lengths, nplanted = bw.make_targets(ntgt, len(hmms), templates, lib_lengths, planted_frac=min(0.5, 12.5 * len(hmms) / ntgt))
db = plan7.SequenceDatabase.from_packed(hmms[0].alphabet, flat, offsets, lengths, device=0)
list(hmmer.hmmsearch(oms, db, feeders=feeders))
for rep in range(2):
    t0 = time.perf_counter()
    hits = list(hmmer.hmmsearch(oms, db, feeders=feeders))
    dt = time.perf_counter() - t0
    st = hmmer.pipeline_stats()
    cells = float(sum(h.M for h in hmms)) * float(lengths.sum())
    w = st.get("wall", dt)
    print(json.dumps({"profiles": nprof, "targets": ntgt, "seconds": round(dt, 3), "gcups": round(cells / dt / 1e9, 1), "ms_per_profile": round(1e3 * dt / nprof, 4),
                      "stats": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()},
                      "batch_ms": {k: round(sum(h.timings_ms[k] for h in hits) / len(hits), 2) for k in ("stage1", "stage2", "host_stage_busy", "batch_queries")}}), flush=True)
