"""The headline workload (KR x 1M synthetic 300-aa targets), one batch of B queries at a time with nothing else in
flight: host time of enqueue / wait / finish and the device stage times, to tell device-bound from host-bound.
usage: config1_phases.py B nbatches"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from pyhmmer_amd import plan7
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
with plan7.HMMFile(os.path.join(ROOT, "tests", "golden", "hmms", "KR.hmm")) as hf:
    hmm = next(iter(hf))
bg = plan7.Background(hmm.alphabet)
om = plan7.OptimizedProfile(hmm, bg, 300)
flat, offsets, lengths, planted = bench.make_workload(hmm, 1_000_000, 300, seed=42)
db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, offsets, lengths, device=0)
pli = plan7.Pipeline(hmm.alphabet)
for it in range(nb + 3):
    t0 = time.perf_counter(); pend = pli._search_enqueue_batch([om] * B, db)
    t1 = time.perf_counter(); plan7.Pipeline._search_wait(pend)
    t2 = time.perf_counter(); hits = plan7.Pipeline._search_finish_batch(pend)
    t3 = time.perf_counter()
    if it >= 3:
        print(f"batch of {B}: enqueue {1e3 * (t1 - t0):.3f}  wait {1e3 * (t2 - t1):.3f}  finish {1e3 * (t3 - t2):.3f} ms; "
              f"device ms { {k: round(v, 3) for k, v in hits[0].timings_ms.items()} }", flush=True)
