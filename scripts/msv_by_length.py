"""MSV kernel rate by model length: one synthetic profile of each length against 500k synthetic 300-aa targets, alone on
the device (batch of 1, nothing else in flight).  usage: msv_by_length.py [M ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, bench_workloads as bw
from pyhmmer_amd import plan7, hmmer
Ms = [int(x) for x in sys.argv[1:]] or [100, 262, 470, 480, 600, 900, 1200, 2000]
templates = bw.load_templates()
bg = plan7.Background(templates[0].alphabet)
kr = [t for t in templates if t.name == "KR"][0] if any(t.name == "KR" for t in templates) else templates[0]
flat, offsets, lengths, planted = bench.make_workload(kr, 500_000, 300, seed=42)
db = plan7.SequenceDatabase.from_packed(kr.alphabet, flat, offsets, lengths, device=0)
res = float(lengths.sum())
for M in Ms:
    hmm = bw.make_entry(templates, 3, M)
    om = plan7.OptimizedProfile(hmm, bg, 300)
    ms = []
    for h in hmmer.hmmsearch((om for _ in range(4)), db, pipeline_depth=0, batch=1):
        ms.append(h.timings_ms["msv_kernel"])
    t = min(ms[1:])
    print(f"M {M:5d}: msv kernel {t:8.3f} ms = {M * res / t / 1e6:8.1f} GCUPS; stage1 {h.timings_ms['stage1']:.2f} ms, past msv {h.stage_counts['msv']}", flush=True)
