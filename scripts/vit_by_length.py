"""Packed Viterbi filter rate by model length: one synthetic profile of each length against 500k synthetic 300-aa targets, alone
on the device (batch of 1, nothing else in flight); cells = survivors of MSV + bias x L x M.  usage: vit_by_length.py [M ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, bench_workloads as bw
from pyhmmer_amd import plan7, hmmer
Ms = [int(x) for x in sys.argv[1:]] or [30, 60, 100, 128, 160, 200, 262, 320, 480, 640]
templates = bw.load_templates()
bg = plan7.Background(templates[0].alphabet)
kr = [t for t in templates if t.name == "KR"][0] if any(t.name == "KR" for t in templates) else templates[0]
flat, offsets, lengths, planted = bench.make_workload(kr, 500_000, 300, seed=42)
db = plan7.SequenceDatabase.from_packed(kr.alphabet, flat, offsets, lengths, device=0)
for M in Ms:
    hmm = bw.make_entry(templates, 3, M)
    om = plan7.OptimizedProfile(hmm, bg, 300)
    ms, sc = [], None
    for h in hmmer.hmmsearch((om for _ in range(4)), db, pipeline_depth=0, batch=1):
        ms.append(h.timings_ms["viterbi"]); sc = h.stage_counts
    t = min(ms[1:])
    cells = sc["bias"] * 300.0 * M
    print(f"M {M:5d}: viterbi stage {t:8.3f} ms for {sc['bias']} targets = {cells / t / 1e6:8.1f} GCUPS ({1e6 * t / max(1, sc['bias']) / 300:.2f} ns per target-row)", flush=True)
