import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import load_hmms, golden_table, GOLDEN
from pyhmmer_amd import plan7, easel
name = sys.argv[1] if len(sys.argv) > 1 else "PF02826"
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    block = sf.read_block()
db = plan7.SequenceDatabase(block)
for hmm in load_hmms(name):
    hits = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, db)
    print("##", hmm.name, len(hits), hits.stage_counts, hits.timings_ms)
    rows = {r[0]: r for r in golden_table(f"{name}.tbl", hmm.name)}
    drows = {}
    for r in golden_table(f"{name}.domtbl", hmm.name, kind="domtbl"):
        drows.setdefault(r[0], []).append(r)
    for h in hits:
        g = rows.get(h.name)
        print(f"{h.name:32s} sc {h.score:7.2f} bias {h.bias:5.2f} E {h.evalue:9.2g} exp {h.nexpected:4.1f} {h.nregions} {h.nclustered} {h.noverlaps} {h.nenvelopes} {len(h.domains)} | gold",
              (g[5], g[6], g[4], g[10:16]) if g else None)
        for d, gd in zip(h.domains, drows.get(h.name, [None] * 10) + [None] * 10):
            print(f"      dom sc {d.score:7.2f} bias {d.bias:5.2f} cE {d.c_evalue:9.2g} hmm {d.alignment.hmm_from}-{d.alignment.hmm_to} ali {d.alignment.target_from}-{d.alignment.target_to} env {d.env_from}-{d.env_to} acc {d.accuracy:.2f} rep {d.reported}| gold",
                  (gd[13], gd[14], gd[11], gd[15:22]) if gd else None)
