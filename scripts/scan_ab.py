"""A/B of a debug option on the resident scan pass (configs[2]): scan_ab.py <option> <value_a> <value_b> [passes]
Builds the 20,000-profile library once, then alternates passes with the option at the two values."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN
import bench_workloads as bw
from pyhmmer_amd import _lib, easel, plan7, hmmer
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    proteome = sf.read_block()
hmms, lengths, templates = bw.make_library(20000, count=20000)
bg = plan7.Background(proteome.alphabet)
block = plan7.OptimizedProfileBlock(proteome.alphabet, (plan7.OptimizedProfile(h, bg, 400) for h in hmms))
cells = float(lengths.sum()) * proteome.total_length()
list(hmmer.hmmscan(proteome, block))                     # device images
ref = None
times = {va: [], vb: []}
for p in range(passes):
    for v in (va, vb):
        _lib.set_debug_option(opt, v)
        t0 = time.perf_counter()
        res = list(hmmer.hmmscan(proteome, block))
        dt = time.perf_counter() - t0
        times[v].append(dt)
        sig = [(h.name, round(h.score, 2), len(h.domains)) for r in res for h in r]
        if ref is None:
            ref = sig
        assert sig == ref, "the two settings give different hits"
for v in (va, vb):
    t = sorted(times[v])
    print(f"{opt}={v}: best {t[0]:.3f} s = {cells / t[0] / 1e9:.0f} GCUPS, median {t[len(t) // 2]:.3f} s, runs {' '.join('%.3f' % x for x in times[v])}", flush=True)
print("stats of the last pass:", hmmer.pipeline_stats())
