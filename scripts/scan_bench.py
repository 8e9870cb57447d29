"""hmmscan orientation timing: the fixture proteome (2,100 sequences) against N profiles, for several feeder counts."""
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
oms = [plan7.OptimizedProfile(h, bg, 400) for h in models] * int(sys.argv[1]) if len(sys.argv) > 1 else 10
cells = sum(om.M for om in oms) * block.total_length()
list(hmmer.hmmscan(block, oms[:14]))          # warm-up: device images of all profiles, kernels
for feeders, depth in ((1, 2), (2, 4), (4, 8), (8, 16), (16, 32)):
    t0 = time.perf_counter()
    res = list(hmmer.hmmscan(block, oms, feeders=feeders, pipeline_depth=depth))
    dt = time.perf_counter() - t0
    print(f"feeders {feeders:2d} depth {depth:2d}: {len(oms)} profiles x {len(block)} seqs in {dt:6.3f} s = {1e3 * dt / len(oms):6.3f} ms/profile, {cells / dt / 1e9:8.1f} GCUPS, hits {sum(len(r) for r in res)}")
for feeders, depth in ((1, 2), (4, 8)):
    acc = {}
    t0 = time.perf_counter()
    n = 0
    for h in hmmer.hmmsearch(oms, block, feeders=feeders, pipeline_depth=depth):
        n += 1
        for k, v in h.timings_ms.items():
            acc[k] = acc.get(k, 0.0) + v
    dt = time.perf_counter() - t0
    print(f"hmmsearch feeders {feeders}: {1e3 * dt / n:.3f} ms/profile;", {k: round(v / n, 3) for k, v in acc.items()})
