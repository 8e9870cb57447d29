"""Envelope kernel time by model length: a synthetic profile of each length against 20,000 targets of 2,200 residues,
1,000 of them with a planted domain (emitted from the profile); one query alone on the device.
usage: env_by_length.py [M ...]   (ENV_PLANTED=fraction of targets with a planted domain, default 0.05)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, bench_workloads as bw
from pyhmmer_amd import plan7, hmmer
frac = float(os.environ.get("ENV_PLANTED", "0.05"))
Ms = [int(x) for x in sys.argv[1:]] or [262, 500, 600, 768, 1000, 1280, 1536, 2000]
templates = bw.load_templates()
bg = plan7.Background(templates[0].alphabet)
for M in Ms:
    hmm = bw.make_entry(templates, 3, M)
    flat, offsets, lengths, planted = bench.make_workload(hmm, 20_000, 2200, seed=7, planted_frac=frac)
    db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, offsets, lengths, device=0)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    ms = []
    for h in hmmer.hmmsearch((om for _ in range(4)), db, pipeline_depth=0, batch=1):
        ms.append((h.timings_ms["envelopes"], h.timings_ms["stage2"], h.timings_ms["host_multi"]))
    e, s2, mu = min(ms[1:])
    nd = sum(len(x.domains) for x in h)
    print(f"M {M:5d}: envelopes {e:8.2f} ms, stage 2 {s2:8.2f} ms (host ensembles {mu:6.2f}); hits {len(h)}, domains {nd}, past fwd {h.stage_counts['fwd']}", flush=True)
