"""The line's workload (Pfam-shaped library x 500,000 targets) one device batch at a time with nothing else in flight
(hmmer.hmmsearch, pipeline_depth = 0): the batches' stage times, and -- for the PMC passes of scripts/pfam_pmc.sh -- the
algorithmic bytes of the fast MSV launches the run made (L + 2 bytes read and 16 bytes written per
comparison, a profile's MSV table once per launch).
usage: pfam_phases.py [profiles=600] [targets=500000]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench_workloads as bw
from pyhmmer_amd import hmmer, plan7

nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ntgt = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
hmms, lib_lengths, templates = bw.make_library(20000, device=0, count=nprof)
bg = plan7.Background(hmms[0].alphabet)
oms = [plan7.OptimizedProfile(h, bg, 400) for h in hmms]
flat, offsets, lengths, nplanted = bw.make_targets(ntgt, len(hmms), templates, lib_lengths, planted_frac=min(0.5, 12.5 * len(hmms) / ntgt))
db = plan7.SequenceDatabase.from_packed(hmms[0].alphabet, flat, offsets, lengths, device=0)
list(hmmer.hmmsearch(oms[:64], db, pipeline_depth=0))          # images, pools
t0 = time.perf_counter()
hits = list(hmmer.hmmsearch(oms, db, pipeline_depth=0))
dt = time.perf_counter() - t0
res, n = float(lengths.sum()), float(len(lengths))
lane = [h for h in hmms if h.M <= 2048] + [h for h in hmms[:64] if h.M <= 2048]      # the models the lane-per-target MSV kernels serve (K = 1, 2, 4, 8 tiles), the warm-up pass included
alg = sum(res + 2.0 * n + 16.0 * n + 29 * 16 * max(2, (h.M - 1) // 16 + 1) for h in lane)
w = np.array([1.0 / max(1.0, h.timings_ms["batch_queries"]) for h in hits])
print(json.dumps({"profiles": nprof, "targets": ntgt, "seconds": round(dt, 3), "gcups": round(sum(h.M for h in hmms) * res / dt / 1e9, 1),
                  "batches": round(float(w.sum()), 1), "lane_kernel_profiles": len(lane),
                  "algorithmic_bytes_of_the_fast_msv_launches": int(alg),
                  "mean_batch_ms": {k: round(float((w * np.array([h.timings_ms[k] for h in hits])).sum() / w.sum()), 3) for k in ("msv_kernel", "msv", "bias", "viterbi", "forward", "fwd_rows", "stage1", "stage2")}}))
