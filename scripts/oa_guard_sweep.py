"""How wide must the optimal-accuracy near-tie guard be?  For a range of cfg.oa_guard: envelopes the host twin had to
repeat, and domains whose integer outputs (coordinates, alignment strings, posterior line) still differ from the host
twin's.  Workloads: config-2 shape with 2 % planted domains (KR), the fixture proteome with five fixture models, and
long random models (envelope kernel instantiations with many nodes per lane)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
from conftest import GOLDEN, load_hmms, random_hmm
from test_gpu_envelopes import _records
from test_gpu_filters import _model_block
from pyhmmer_amd import easel, plan7

cases = []
hmm = load_hmms("KR")[0]
flat, off, ln, planted = bench.make_workload(hmm, 150_000, 300, 7, planted_frac=0.02)
cases.append(("KR planted", hmm, plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, off, ln), {}))
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=hmm.alphabet) as sf:
    prot = plan7.SequenceDatabase(sf.read_block())
for name in ("PF02826", "Thioesterase", "LuxC"):
    cases.append((name, load_hmms(name)[0], prot, dict(E=1e3, domE=1e3)))
for M in (640, 1100):
    h = random_hmm(M, seed=3000 + M)
    cases.append((f"random M={M}", h, plan7.SequenceDatabase(_model_block(h, 300, 400, seed=M)), dict(E=1e3, domE=1e3)))
for label, hmm, db, opts in cases:
    host = {r[0]: r for r in _records(plan7.Pipeline(hmm.alphabet, host_envelopes=True, host_regions=True, **opts).search_hmm(hmm, db))}
    for g in (0.0, 2.5e-7, 1e-6, 2e-6, 8e-6, 3e-5):
        hits = plan7.Pipeline(hmm.alphabet, oa_guard=g, **opts).search_hmm(hmm, db)
        dev = {r[0]: r for r in _records(hits)}
        ndom = sum(len(r[2]) for r in dev.values())
        bad = sum(1 for k, r in dev.items() for (ia, fa), (ib, fb) in zip(r[2], host[k][2]) if ia != ib) if dev.keys() == host.keys() else -1
        print(f"{label:16s} guard {g:8.1e}: domains {ndom:6d} redone {hits.guard_counts['oa_redone']:5d} differing {bad}", flush=True)
