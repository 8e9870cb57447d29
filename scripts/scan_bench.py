"""hmmscan orientation timing: the fixture proteome (2,100 sequences) against N x 14 profiles (every profile its own
OptimizedProfile object, so each one pays for its device image), for several batch / feeder / window settings (every
setting twice: the first pass creates the workspaces and buffers of that shape, the second is the steady state), and the
host time of the three phases of one batch (enqueue / wait / finish) on a single thread.
usage: scan_bench.py [N] [batch,feeders,depth,window ...]"""
import sys, time
import os
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def fresh():
    t0 = time.perf_counter()
    oms = [plan7.OptimizedProfile(h, bg, 400) for _ in range(reps) for h in models]
    return oms, time.perf_counter() - t0


oms, tb = fresh()
print(f"{len(oms)} profiles built on the host in {1e3 * tb / len(oms):.3f} ms/profile")
cells = sum(om.M for om in oms) * block.total_length()
list(hmmer.hmmscan(block, oms[:28]))          # warm-up: kernels, workspaces
configs = ((1, 4, 32, 4), (64, 3, 3, 1), (256, 2, 2, 1), (256, 3, 3, 1), (256, 4, 4, 1))
if len(sys.argv) > 2:
    configs = tuple(tuple(int(x) for x in c.split(",")) for c in sys.argv[2:])
for batch, feeders, depth, window in configs:
    # first pass: workspaces, envelope buffers and pinned blocks of this (batch, feeders) shape are created on the way
    # (cold); second pass: the same shape with fresh profiles (every profile still pays for its device image)
    for label in ("cold", "warm"):
        oms, _ = fresh()
        t0 = time.perf_counter()
        res = list(hmmer.hmmscan(block, oms, feeders=feeders, pipeline_depth=depth, window=window, batch=batch))
        dt = time.perf_counter() - t0
        print(f"batch {batch:3d} feeders {feeders:2d} depth {depth:2d} window {window:2d} {label}: {len(oms)} profiles x {len(block)} seqs in {dt:6.3f} s = "
              f"{1e3 * dt / len(oms):6.3f} ms/profile, {cells / dt / 1e9:8.1f} GCUPS, hits {sum(len(r) for r in res)}", flush=True)
# resident images (second pass over the same OptimizedProfile objects)
t0 = time.perf_counter()
res = list(hmmer.hmmscan(block, oms, batch=64))
dt = time.perf_counter() - t0
print(f"batch 64, device images already resident: {1e3 * dt / len(oms):6.3f} ms/profile, {cells / dt / 1e9:8.1f} GCUPS")
# phases of one batch on one thread
db = plan7.SequenceDatabase(block)
pli = plan7.Pipeline(block.alphabet)
for B in (1, 16, 64, 256):
    acc = [0.0, 0.0, 0.0]
    nb = 0
    for lo in range(0, min(len(oms), max(8 * B, 512)), B):
        qs = oms[lo:lo + B]
        t0 = time.perf_counter(); pend = pli._search_enqueue_batch(qs, db)
        t1 = time.perf_counter(); plan7.Pipeline._search_wait(pend)
        t2 = time.perf_counter(); plan7.Pipeline._search_finish_batch(pend)
        t3 = time.perf_counter()
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; nb += len(qs)
    print(f"one thread, batch {B:3d}: enqueue {1e3 * acc[0] / nb:.3f}  wait {1e3 * acc[1] / nb:.3f}  finish {1e3 * acc[2] / nb:.3f} ms per profile")
