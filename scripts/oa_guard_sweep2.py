"""Finer sweep of cfg.oa_guard on many more domains (see oa_guard_sweep.py): which is the smallest guard that leaves no
domain different from the host twin?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
from conftest import load_hmms, random_hmm
from test_gpu_filters import _model_block
from oa_guard_sweep import _records
from pyhmmer_amd import plan7

GUARDS = (0.0, 1.25e-7, 2.5e-7, 5e-7, 1e-6, 2e-6, 3e-6)
tot = {g: [0, 0, 0] for g in GUARDS}
cases = []
kr = load_hmms("KR")[0]
for seed in range(4):
    flat, off, ln, planted = bench.make_workload(kr, 200_000, 300, 100 + seed, planted_frac=0.02)
    cases.append((f"KR seed {seed}", kr, plan7.SequenceDatabase.from_packed(kr.alphabet, flat, off, ln), {}))
for M in (100, 200, 320, 450, 640, 900, 1100, 1500, 2000):
    for seed in (1, 2):
        h = random_hmm(M, seed=7000 + 10 * M + seed)
        cases.append((f"random M={M} s{seed}", h, plan7.SequenceDatabase(_model_block(h, 300, 600, seed=M + seed)), dict(E=1e3, domE=1e3)))
for label, hmm, db, opts in cases:
    host = _records(plan7.Pipeline(hmm.alphabet, host_envelopes=True, host_regions=True, **opts).search_hmm(hmm, db))
    line = []
    for g in GUARDS:
        hits = plan7.Pipeline(hmm.alphabet, oa_guard=g, **opts).search_hmm(hmm, db)
        dev = _records(hits)
        ndom = sum(len(r) for r in dev.values())
        bad = sum(1 for k, r in dev.items() for ia, ib in zip(r, host[k]) if ia != ib) if dev.keys() == host.keys() else -1
        tot[g][0] += ndom; tot[g][1] += hits.guard_counts["oa_redone"]; tot[g][2] += bad
        line.append(f"{g:.2g}:{hits.guard_counts['oa_redone']}/{bad}")
    print(f"{label:20s} domains {ndom:6d}  redone/differing  " + "  ".join(line), flush=True)
for g in GUARDS:
    print(f"TOTAL guard {g:8.2e}: domains {tot[g][0]} redone {tot[g][1]} ({100.0 * tot[g][1] / max(1, tot[g][0]):.2f} %) differing {tot[g][2]}")
