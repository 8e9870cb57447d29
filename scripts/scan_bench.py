"""hmmscan orientation timing: the fixture proteome (2,100 sequences) against N x 14 profiles (every profile its own
OptimizedProfile object, so each one pays for its device image), for several feeder / window settings."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
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
list(hmmer.hmmscan(block, oms[:14]))          # warm-up: kernels, workspaces
for feeders, depth, window in ((1, 2, 1), (4, 8, 1), (1, 8, 8), (2, 32, 8), (2, 64, 16), (4, 64, 8), (1, 32, 32)):
    oms, _ = fresh()
    t0 = time.perf_counter()
    res = list(hmmer.hmmscan(block, oms, feeders=feeders, pipeline_depth=depth, window=window))
    dt = time.perf_counter() - t0
    print(f"feeders {feeders:2d} depth {depth:2d} window {window:2d}: {len(oms)} profiles x {len(block)} seqs in {dt:6.3f} s = "
          f"{1e3 * dt / len(oms):6.3f} ms/profile, {cells / dt / 1e9:8.1f} GCUPS, hits {sum(len(r) for r in res)}", flush=True)
acc = {}
n = 0
oms, _ = fresh()
t0 = time.perf_counter()
for h in hmmer.hmmsearch(oms, block, feeders=1, pipeline_depth=2):
    n += 1
    for k, v in h.timings_ms.items():
        acc[k] = acc.get(k, 0.0) + v
dt = time.perf_counter() - t0
print(f"hmmsearch (one feeder): {1e3 * dt / n:.3f} ms/profile;", {k: round(v / n, 3) for k, v in acc.items()})
