"""Envelope stage of the headline workload alone on the device (no overlap with other batches): KR x 1e6 x 300 aa,
batches of 7 queries, pipeline_depth 0.  Prints the envelope stage's wall time per batch."""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import bench
from pyhmmer_amd import plan7, hmmer
with plan7.HMMFile(ROOT / "tests" / "golden" / "hmms" / "KR.hmm") as hf:
    hmm = next(iter(hf))
bg = plan7.Background(hmm.alphabet)
om = plan7.OptimizedProfile(hmm, bg, 300)
flat, offsets, lengths, planted = bench.make_workload(hmm, 1_000_000, 300, seed=42)
db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, offsets, lengths, device=0)
depth = int(os.environ.get("DEPTH", "0"))
out = []
for h in hmmer.hmmsearch((om for _ in range(42)), db, pipeline_depth=depth, batch=7):
    out.append(h.timings_ms["envelopes"])
print({k: round(v, 2) for k, v in h.timings_ms.items()})
print(sys.argv[1] if len(sys.argv) > 1 else "", "envelope stage per batch of 7 (ms):", " ".join(f"{x:.1f}" for x in out[::7]))
