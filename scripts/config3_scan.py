"""BASELINE configs[2] at Pfam size: the synthetic 20,000-entry profile library (bench_workloads.py) scanned against the
2,100-sequence fixture proteome with hmmer.hmmscan (defaults).  First pass: every profile pays for its device image;
second pass: images resident.  usage: config3_scan.py [n_profiles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN
import bench_workloads as bw
from pyhmmer_amd import easel, plan7, hmmer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    proteome = sf.read_block()
t0 = time.perf_counter()
hmms, lengths, templates = bw.make_library(n, count=n)
bg = plan7.Background(proteome.alphabet)
block = plan7.OptimizedProfileBlock(proteome.alphabet, (plan7.OptimizedProfile(h, bg, 400) for h in hmms))
print(f"{n} profiles (M {int(lengths.min())}..{int(lengths.max())}, mean {lengths.mean():.0f}) built in {time.perf_counter() - t0:.1f} s")
cells = float(lengths.sum()) * proteome.total_length()
list(hmmer.hmmscan(proteome, block[:64]))
for label in ("device images built on the way", "device images resident"):
    t0 = time.perf_counter()
    res = list(hmmer.hmmscan(proteome, block))
    dt = time.perf_counter() - t0
    print(f"hmmscan, {label}: {dt:.3f} s = {1e3 * dt / n:.4f} ms per profile, {cells / dt / 1e9:.0f} GCUPS, {len(proteome) / dt:.0f} query sequences/s, hits {sum(len(r) for r in res)}", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "trace":          # where a resident pass spends its time
    import json
    print("pipeline_stats of the last pass:", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in hmmer.pipeline_stats().items()}), flush=True)
    from pyhmmer_amd import _lib
    L = _lib.lib()
    def timed(name):            # wall time of a C entry point, on stderr (the CDLL instance caches its functions as attributes)
        fn = getattr(L, name)
        def call(*a):
            t = time.perf_counter(); r = fn(*a)
            print(f"[py] {name} {1e3 * (time.perf_counter() - t):.2f} ms", file=sys.stderr, flush=True)
            return r
        setattr(L, name, call)
    for name in ("p7x_search_batch_finish", "p7x_scan_accum_add_indexed", "p7x_scan_accum_finish", "p7x_search_batch_enqueue"):
        timed(name)
    _lib.set_debug_option("trace_finish", 1)
    hmmer.PIPE_TRACE = True
    t0 = time.perf_counter()
    res = list(hmmer.hmmscan(proteome, block))
    print(f"traced pass: {time.perf_counter() - t0:.3f} s", flush=True)
