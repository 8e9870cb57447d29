"""Where a batch of the hmmscan orientation spends its time: device phases of one result per batch (event timers of the
cascade) and, with P7X_FINISH_DEBUG=1, the wall time of the phases of the host stage (stderr).
usage: scan_phases.py [reps] [batch] [feeders]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_hmms, GOLDEN
from pyhmmer_amd import easel, plan7, hmmer
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    block = sf.read_block()
models = []
for name in ("PF02826", "Thioesterase", "RREFam", "KR", "LuxC"):
    models += load_hmms(name)
bg = plan7.Background(models[0].alphabet)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
feeders = int(sys.argv[3]) if len(sys.argv) > 3 else 3
list(hmmer.hmmscan(block, [plan7.OptimizedProfile(h, bg, 400) for h in models]))
for label in ("cold", "warm"):
    oms = [plan7.OptimizedProfile(h, bg, 400) for _ in range(reps) for h in models]
    sys.stderr.write(f"==== {label}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    res = list(hmmer.hmmscan(block, oms, feeders=feeders, pipeline_depth=feeders, window=1, batch=batch))
    dt = time.perf_counter() - t0
    print(f"batch {batch} feeders {feeders} {label}: {1e3 * dt / len(oms):.3f} ms/profile", flush=True)
