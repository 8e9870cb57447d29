"""hmmscan orientation (20k-profile library x fixture proteome, images resident): sweep of batch size, feeders, depth."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN
import bench_workloads as bw
from pyhmmer_amd import easel, plan7, hmmer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    proteome = sf.read_block()
hmms, lengths, templates = bw.make_library(n, count=n)
bg = plan7.Background(proteome.alphabet)
block = plan7.OptimizedProfileBlock(proteome.alphabet, (plan7.OptimizedProfile(h, bg, 400) for h in hmms))
cells = float(lengths.sum()) * proteome.total_length()
list(hmmer.hmmscan(proteome, block))
for batch, feeders, depth, window in [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]] or [(256, 3, 3, 1), (1024, 3, 3, 1), (2048, 2, 3, 1)]:
    runs = []
    for rep in range(4):
        t0 = time.perf_counter()
        res = list(hmmer.hmmscan(proteome, block, batch=batch, feeders=feeders, pipeline_depth=depth, window=window))
        runs.append(time.perf_counter() - t0)
    best = min(runs)
    print(f"batch {batch:5d} feeders {feeders} depth {depth} window {window}: {best:.3f} s = {cells / best / 1e9:.0f} GCUPS (runs {' '.join(f'{r:.3f}' for r in runs)})", flush=True)
