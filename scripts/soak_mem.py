"""Soak: the line's workload (8,000 library profiles x 500,000 targets) three times in one process; device memory in use and host RSS after
each pass (grow-only pools must settle after the first), hit lists identical between passes."""
import os, sys, time, json, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_workloads as bw
from pyhmmer_amd import plan7, hmmer
nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
hmms, lib_lengths, templates = bw.make_library(20000, device=0, count=nprof)
bg = plan7.Background(hmms[0].alphabet)
oms = [plan7.OptimizedProfile(h, bg, 400) for h in hmms]
flat, offsets, lengths, nplanted = bw.make_targets(500000, len(hmms), templates, lib_lengths, planted_frac=min(0.5, 12.5 * len(hmms) / 500000))
db = plan7.SequenceDatabase.from_packed(hmms[0].alphabet, flat, offsets, lengths, device=0)
ref = None
for rep in range(3):
    t0 = time.perf_counter()
    hits = list(hmmer.hmmsearch(oms, db))
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    sig = [(len(h), round(sum(x.score for x in h), 3)) for h in hits]
    if ref is None: ref = sig
    print(json.dumps({"pass": rep, "seconds": round(dt, 3), "device_used_gb": round((total - free) / 2**30, 2),
                      "host_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 2), "hits": sum(s[0] for s in sig), "same_as_first": sig == ref}), flush=True)
