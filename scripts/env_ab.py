"""A/B: envelope rescoring on the device (default) vs on the host (P7X_HOST_ENVELOPES=1).
Prints the differences in hits / domain coordinates / alignment strings / scores."""
import os, pickle, subprocess, sys
CODE = r'''
import sys, pickle
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bench
from conftest import load_hmms, GOLDEN
from pyhmmer_amd import plan7, easel
out = {}
def grab(tag, hits):
    rec = []
    for h in hits:
        doms = []
        for d in h.domains:
            a = d.alignment
            doms.append((d.env_from, d.env_to, a.target_from, a.target_to, a.hmm_from, a.hmm_to, d.score, d.bias, d.accuracy,
                         a.target_sequence, a.hmm_sequence, a.identity_sequence, a.posterior_probabilities))
        rec.append((h.name, h.score, h.bias, h.evalue, doms))
    out[tag] = (rec, hits.timings_ms, hits.stage_counts)
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    block = sf.read_block()
db = plan7.SequenceDatabase(block)
for name in ("PF02826", "Thioesterase", "RREFam", "KR", "LuxC"):
    for hmm in load_hmms(name):
        grab(name + ":" + hmm.name, plan7.Pipeline(hmm.alphabet, E=1e3, domE=1e3).search_hmm(hmm, db))
hmm = load_hmms("KR")[0]
flat, off, ln, planted = bench.make_workload(hmm, int(sys.argv[2]), 300, 42)
names = [b"t%d" % i for i in range(len(ln))]
db2 = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, off, ln)
pl = plan7.Pipeline(hmm.alphabet)
for rep in range(3):
    hits = pl.search_hmm(hmm, db2)
grab("bench", hits)
pickle.dump(out, open(sys.argv[1], "wb"))
'''
nseq = sys.argv[1] if len(sys.argv) > 1 else "1000000"
for tag, env in (("dev", {}), ("host", {"P7X_HOST_ENVELOPES": "1"})):
    subprocess.run([sys.executable, "-c", CODE, f"/tmp/envab_{tag}.pkl", nseq], check=True, env={**os.environ, **env})
A = pickle.load(open("/tmp/envab_dev.pkl", "rb")); B = pickle.load(open("/tmp/envab_host.pkl", "rb"))
for key in A:
    (ra, ta, sa), (rb, tb, sb) = A[key], B[key]
    nd = sum(len(h[4]) for h in rb)
    bad = []
    if len(ra) != len(rb): bad.append(f"nhits {len(ra)} vs {len(rb)}")
    dmax = 0.0
    for ha, hb in zip(ra, rb):
        if ha[0] != hb[0]: bad.append(f"name {ha[0]} vs {hb[0]}"); continue
        dmax = max(dmax, abs(ha[1] - hb[1]), abs(ha[2] - hb[2]))
        if len(ha[4]) != len(hb[4]): bad.append(f"{ha[0]}: ndom {len(ha[4])} vs {len(hb[4])}"); continue
        for da, dbb in zip(ha[4], hb[4]):
            if da[:6] != dbb[:6]: bad.append(f"{ha[0]}: coords {da[:6]} vs {dbb[:6]}")
            elif da[9:12] != dbb[9:12]: bad.append(f"{ha[0]}: alignment strings differ")
            elif da[12] != dbb[12]: bad.append(f"{ha[0]}: pp line {da[12]} vs {dbb[12]}")
            dmax = max(dmax, abs(da[6] - dbb[6]), abs(da[7] - dbb[7]), abs(da[8] - dbb[8]))
    print(f"{key:40s} hits {len(rb):5d} doms {nd:5d} max|dscore| {dmax:.2e} problems {len(bad)}", bad[:4])
    if key == "bench":
        print("   dev timings", {k: round(v, 3) for k, v in ta.items()})
        print("   host timings", {k: round(v, 3) for k, v in tb.items()})
