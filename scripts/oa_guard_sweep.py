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
from test_gpu_filters import _model_block
from pyhmmer_amd import easel, plan7

def _records(hits):
    out = {}
    for h in hits:
        out[(h.name, h.seqidx)] = [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to,
                                    d.alignment.target_sequence, d.alignment.hmm_sequence, d.alignment.identity_sequence,
                                    d.alignment.posterior_probabilities) for d in h.domains]
    return out


if __name__ == "__main__":
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
      host = _records(plan7.Pipeline(hmm.alphabet, host_envelopes=True, host_regions=True, **opts).search_hmm(hmm, db))
      for g in (0.0, 2.5e-7, 2e-6, 3e-5):
          hits = plan7.Pipeline(hmm.alphabet, oa_guard=g, **opts).search_hmm(hmm, db)
          dev = _records(hits)
          ndom = sum(len(r) for r in dev.values())
          bad = sum(1 for k, r in dev.items() for ia, ib in zip(r, host[k]) if ia != ib) if dev.keys() == host.keys() else -1
          gc = hits.guard_counts
          print(f"{label:16s} guard {g:8.1e}: domains {ndom:6d} redone {gc['oa_redone']:5d} differing {bad}  why {gc['oa_why']}", flush=True)

