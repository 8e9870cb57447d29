"""nhmmer: seconds per search of a stream of the same query by the number of searches in flight (bmyD x 250 Mbp x 2 strands).
usage: nh_inflight.py [k ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw
from pyhmmer_amd import plan7, hmmer, easel
with plan7.HMMFile(os.path.join(ROOT, "tests", "golden", "hmms", "bmyD.hmm")) as hf:
    hmm = next(iter(hf))
seq = bw.make_chromosome(hmm, int(250e6), planted=50, seed=45)
block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chrSyn0", sequence=seq)])
list(hmmer.nhmmer([hmm] * 2, block, devices=[0]))
for k in [int(x) for x in sys.argv[1:]] or [4, 6, 8]:
    for rep in range(2):
        t0 = time.perf_counter()
        n = 12
        hits = list(hmmer.nhmmer([hmm] * n, block, devices=[0], searches_in_flight=k))
        dt = time.perf_counter() - t0
        print(f"in flight {k}: {dt / n:.4f} s per search ({len(hits[-1])} hits)", flush=True)
