"""nhmmer on a hit-rich target (a repeat family: thousands of copies of the model): where the envelopes are rescored.
usage: nhmmer_hitrich.py [Mbp] [copies]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workloads as bw
from pyhmmer_amd import easel, plan7, hmmer
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
with plan7.HMMFile(os.path.join(ROOT, "tests", "golden", "hmms", "bmyD.hmm")) as f:
    hmm = next(iter(f))
seq = bw.make_chromosome(hmm, int(mbp * 1e6), planted=copies, seed=31)
block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chrR", sequence=seq)])
for where, label in ((1, "host workers"), (2, "envelope kernel"), (0, "default")):
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter()
        hits = next(hmmer.nhmmer(hmm, block, host_envelopes=where))
        best = min(best, time.perf_counter() - t0)
    print(f"{mbp:g} Mbp, {copies} planted copies, envelopes by the {label:16s}: {best:.3f} s, hits {len(hits)}, host tail {hits.timings_ms['host_domaindef']:.0f} ms", flush=True)
