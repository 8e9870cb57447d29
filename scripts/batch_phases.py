"""One thread, batches of B fixture profiles against the fixture proteome: host time of enqueue / wait / finish per
batch after a warm-up, for profiling (rocprofv3 --kernel-trace) the batched cascade.  usage: batch_phases.py B nbatches [scan]"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_hmms, GOLDEN
from pyhmmer_amd import easel, plan7
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 10
scan = len(sys.argv) > 3
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    block = sf.read_block()
models = []
for name in ("PF02826", "Thioesterase", "RREFam", "KR", "LuxC"):
    models += load_hmms(name)
bg = plan7.Background(models[0].alphabet)
oms = [plan7.OptimizedProfile(models[i % len(models)], bg, 400) for i in range(B)]
db = plan7.SequenceDatabase(block)
pli = plan7.Pipeline(block.alphabet)
if scan:
    pli._mode = plan7._P7X_SCAN_MODELS
for it in range(nb + 3):
    t0 = time.perf_counter(); pend = pli._search_enqueue_batch(oms, db)
    t1 = time.perf_counter(); plan7.Pipeline._search_wait(pend)
    t2 = time.perf_counter(); hits = plan7.Pipeline._search_finish_batch(pend)
    t3 = time.perf_counter()
    if it >= 3:
        print(f"batch of {B}: enqueue {1e3 * (t1 - t0):.3f}  wait {1e3 * (t2 - t1):.3f}  finish {1e3 * (t3 - t2):.3f} ms; "
              f"hits {sum(len(h) for h in hits)}; device ms {[round(v, 3) for v in list(hits[0].timings_ms.values())[:5]]}", flush=True)
