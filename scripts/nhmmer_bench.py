"""BASELINE configs[4]: nhmmer long-target, one DNA HMM (fixture bmyD, M = 1203) against a synthetic chromosome (i.i.d.
ACGT 0.25, seed 45; SURVEY.md 8d "config 5"), both strands, block_length 262144.  Reports the SSV scan kernel (HIP
events) as GCUPS = 2 strands x L x M / time, and the whole search.  usage: nhmmer_bench.py [Mbp] [planted] [ssv_kernel option: 3 = row maximum in every row, 4 = in every second row | trace]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pyhmmer_amd import _lib, easel, plan7
if len(sys.argv) > 3 and sys.argv[3] == "trace":
    _lib.set_debug_option("trace_longtarget", 1)
elif len(sys.argv) > 3:
    _lib.set_debug_option("ssv_kernel", int(sys.argv[3]))
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 250.0
planted = int(sys.argv[2]) if len(sys.argv) > 2 else 50
with plan7.HMMFile(os.path.join(ROOT, "tests", "golden", "hmms", "bmyD.hmm")) as f:
    hmm = next(iter(f))
L = int(mbp * 1e6)
import bench_workloads as bw
seq = bw.make_chromosome(hmm, L, planted)
block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chrSyn", sequence=seq)])
pli = plan7.LongTargetsPipeline(hmm.alphabet)
for it in range(2):
    t0 = time.perf_counter()
    hits = pli.search_hmm(hmm, block)
    dt = time.perf_counter() - t0
    scan_ms = hits.timings_ms["msv_kernel"]
    cells = 2.0 * L * hmm.M
    print(f"run {it}: {mbp:g} Mbp x 2 strands x M={hmm.M}: SSV scan kernels {scan_ms:.2f} ms = {cells / scan_ms / 1e6:.0f} GCUPS; "
          f"whole search {dt:.2f} s = {cells / dt / 1e9:.0f} GCUPS; windows past msv/bias/vit/fwd {hits.stage_counts}, "
          f"hits {len(hits)} (checksum {sum(int(h.best_domain.alignment.target_from) * 7 + int(h.best_domain.alignment.hmm_from) for h in hits) % 1000003}) reported {len(hits.reported)} (planted {planted}); ms: scan+seeds wall {hits.timings_ms['msv']:.0f}, window batch on device {hits.timings_ms['bias']:.0f}, host tail {hits.timings_ms['host_domaindef']:.0f}", flush=True)
