"""The product's HOST half (p7_domaindef + p7_tophits restatement in libp7x) against real HMMER output.  No GPU:
the device half is stood in for by the oracle (tests/host_pipeline.py) and the rows go through the exported
C-ABI entry point p7x_postprocess_targets."""
import io
import itertools
import pickle

import pytest

import host_pipeline
from conftest import golden_table
from golden_checks import check_domtbl, check_tbl
from pyhmmer_amd import easel, plan7


def test_host_pf02826_matches_hmmer_tables(models, oracle, proteome):
    hmm = models["PF02826"][0]
    hits = host_pipeline.host_search(oracle, hmm, proteome)
    assert len(hits) == 22 and hits.Z == 2100
    check_tbl(hits, golden_table("PF02826.tbl"))
    check_domtbl(hits, golden_table("PF02826.domtbl", kind="domtbl"))


def test_host_rrefam_matches_hmmer_tables(models, oracle, proteome):
    for hmm in models["RREFam"]:
        hits = host_pipeline.host_search(oracle, hmm, proteome)
        check_tbl(hits, golden_table("RREFam.tbl", hmm.name))
        check_domtbl(hits, golden_table("RREFam.domtbl", hmm.name, kind="domtbl"))


def test_host_thioesterase_alignment(models, oracle, proteome):
    """reference tests/test_hmmer.py:51-106, including the alignment display lines."""
    hmm = models["Thioesterase"][0]
    hits = host_pipeline.host_search(oracle, hmm, proteome)
    assert len(hits) == 1
    hit = hits[0]
    assert hit.name == "938293.PRJEB85.HG003687_113"
    assert hit.score == pytest.approx(8.6, abs=0.1) and hit.bias == pytest.approx(1.5, abs=0.1)
    assert hit.evalue == pytest.approx(0.096, abs=0.01)
    d = hit.domains[0]
    assert d.score == pytest.approx(8.1, abs=0.1) and d.bias == pytest.approx(1.5, abs=0.1)
    assert d.i_evalue == pytest.approx(0.14, abs=0.005) and d.c_evalue == pytest.approx(6.5e-05, abs=2e-6)
    assert (d.alignment.target_from, d.alignment.target_to, d.alignment.target_length) == (115, 129, 261)
    assert (d.alignment.hmm_from, d.alignment.hmm_to, d.alignment.hmm_length) == (79, 93, 243)
    assert (d.env_from, d.env_to) == (115, 129)
    assert d.alignment.hmm_sequence == "GWSfGGvlAyEmArq"
    assert d.alignment.identity_sequence == "G+S+GG +A ++A++"
    assert d.alignment.target_sequence == "GHSMGGSVAVAIAHE"
    assert d.alignment.posterior_probabilities == "9************96"
    assert d.accuracy == pytest.approx(0.96, abs=0.01)


def test_host_merge_equals_whole(models, oracle, proteome):
    """reference tests/test_plan7/test_tophits.py:191-224."""
    hmm = models["PF02826"][0]
    whole = host_pipeline.host_search(oracle, hmm, proteome)
    parts = [host_pipeline.host_search(oracle, hmm, proteome[a:b]) for a, b in ((0, 1000), (1000, 2000), (2000, 2100))]
    merged = parts[0].merge(*parts[1:])
    assert merged.Z == whole.Z == 2100 and merged.domZ == whole.domZ
    assert merged.stage_counts == whole.stage_counts and merged.searched_residues == whole.searched_residues
    assert [h.name for h in merged] == [h.name for h in whole]
    for a, b in zip(merged, whole):
        assert (a.score, a.pre_score, a.sum_score, a.evalue, a.reported, a.included) == \
               (b.score, b.pre_score, b.sum_score, b.evalue, b.reported, b.included)
        for da, db in itertools.zip_longest(a.domains, b.domains):
            assert (da.env_from, da.env_to, da.score, da.c_evalue, da.i_evalue, da.reported, da.included) == \
                   (db.env_from, db.env_to, db.score, db.c_evalue, db.i_evalue, db.reported, db.included)
    # merging pipelines configured differently is refused (plan7.pyx:8832-8861)
    other = host_pipeline.host_search(oracle, hmm, proteome[:500], pipeline=plan7.Pipeline(hmm.alphabet, E=1.0))
    with pytest.raises(ValueError):
        whole.merge(other)


def test_host_thresholds_and_z(models, oracle, proteome):
    hmm = models["PF02826"][0]
    fixed = host_pipeline.host_search(oracle, hmm, proteome, pipeline=plan7.Pipeline(hmm.alphabet, Z=25))
    assert fixed.Z == 25
    auto = host_pipeline.host_search(oracle, hmm, proteome)
    for a, b in zip(fixed, auto):
        assert a.name == b.name and a.evalue == pytest.approx(b.evalue * 25 / 2100, rel=1e-6)
    ga = host_pipeline.host_search(oracle, hmm, proteome, pipeline=plan7.Pipeline(hmm.alphabet, bit_cutoffs="gathering"))
    assert len(ga.reported) == 7 and all(h.score >= 25.1 for h in ga.reported)
    byT = host_pipeline.host_search(oracle, hmm, proteome, pipeline=plan7.Pipeline(hmm.alphabet, T=20.0, domT=10.0))
    assert all(h.score >= 20.0 for h in byT.reported) and len(byT.reported) == 8
    assert all(d.score >= 10.0 for h in byT.reported for d in h.domains.reported)
    nonull2 = host_pipeline.host_search(oracle, hmm, proteome, pipeline=plan7.Pipeline(hmm.alphabet, null2=False))
    assert all(h.bias == pytest.approx(0.0, abs=1e-4) or h.sum_score >= h.pre_score - 1e-3 for h in nonull2)


def _golden_lines(name):
    lines = open(host_pipeline_golden() / "tables" / name).read().splitlines()
    while lines[-1].startswith("#"):
        lines.pop()
    return lines


def host_pipeline_golden():
    from conftest import GOLDEN
    return GOLDEN


@pytest.mark.parametrize("fmt,table", [("targets", "PF02826.tbl"), ("domains", "PF02826.domtbl")])
def test_host_write_reproduces_hmmer_tables_byte_for_byte(models, oracle, proteome, fmt, table):
    """reference tests/test_plan7/test_tophits.py:359-381 (`TopHits.write` against the --tblout / --domtblout text).
    Header and every row are compared as text, sampled-null2 hits included."""
    hmm = models["PF02826"][0]
    hits = host_pipeline.host_search(oracle, hmm, proteome)
    buf = io.BytesIO()
    hits.write(buf, format=fmt)
    got = buf.getvalue().decode().splitlines()
    want = _golden_lines(table)
    assert got == want
    assert sum(1 for h in hits if h.nclustered > 0) == 6          # the stochastic ensemble is exercised
    nohdr = io.BytesIO()
    hits.write(nohdr, format=fmt, header=False)
    assert nohdr.getvalue().decode().splitlines() == got[3:]
    with pytest.raises(ValueError):
        hits.write(io.BytesIO(), format="nonsense")


def test_host_write_pfam_format_holds_the_domain_table_rows(models, oracle, proteome):
    """reference plan7.pyx:9155-9164 (`format="pfam"`, p7_tophits_TabularXfam).  No fixture pins the layout; what is checked is
    its content against the pinned tables: one row per reported hit in hit order, a blank line, one row per reported domain
    carrying the --domtblout numbers, best bit score first, each with the ordinal of the domain inside its hit."""
    hmm = models["PF02826"][0]
    hits = host_pipeline.host_search(oracle, hmm, proteome)
    buf = io.BytesIO()
    hits.write(buf, format="pfam", header=False)
    text = buf.getvalue().decode()
    seqpart, dompart = text.split("\n\n", 1)
    seqrows = [l.split() for l in seqpart.splitlines() if not l.startswith("#")]
    domrows = [l.split() for l in dompart.splitlines() if not l.startswith("#")]
    assert seqpart.startswith("# Sequence scores\n# ---------------\n#\n") and dompart.startswith("# Domain scores\n# -------------\n#\n")
    want_hits = [l.split() for l in _golden_lines("PF02826.tbl") if not l.startswith("#")]
    want_doms = [l.split() for l in _golden_lines("PF02826.domtbl") if not l.startswith("#")]
    assert [r[0] for r in seqrows] == [w[0] for w in want_hits]
    assert [(r[1], r[2], r[3], r[4]) for r in seqrows] == [(w[5], w[4], w[15], w[10]) for w in want_hits]      # bits, E-value, ndom (dom), exp
    assert len(domrows) == len(want_doms)
    scores = [float(r[1]) for r in domrows]
    assert scores == sorted(scores, reverse=True)
    key = lambda name, envst, envto: (name, envst, envto)
    table = {key(w[0], w[19], w[20]): w for w in want_doms}
    for r in domrows:
        w = table[key(r[0], r[5], r[6])]
        # bits, i-Evalue, ordinal, bias, ali from/to, hmm from/to
        assert (r[1], r[2], r[3], r[4], r[7], r[8], r[9], r[10]) == (w[13], w[12], w[9], w[14], w[17], w[18], w[15], w[16])


def test_host_tophits_thresholds(models, oracle, proteome):
    """reference plan7.pyx:8600-8690: the thresholds a hit list was made with; a bit-score threshold is None while its E-value twin
    is in force."""
    hmm = models["PF02826"][0]
    base = host_pipeline.host_search(oracle, hmm, proteome[:400])
    assert (base.E, base.T, base.domE, base.domT, base.incE, base.incT, base.incdomE, base.incdomT, base.bit_cutoffs) == \
           (10.0, None, 10.0, None, 0.01, None, 0.01, None, None)
    byT = host_pipeline.host_search(oracle, hmm, proteome[:400], pipeline=plan7.Pipeline(hmm.alphabet, T=20.0, domT=10.0, incT=25.0, incdomT=12.0))
    assert (byT.T, byT.domT, byT.incT, byT.incdomT) == (20.0, 10.0, 25.0, 12.0)
    ga = host_pipeline.host_search(oracle, hmm, proteome[:400], pipeline=plan7.Pipeline(hmm.alphabet, bit_cutoffs="gathering"))
    assert ga.bit_cutoffs == "gathering"


def test_host_tophits_api(models, oracle, proteome):
    """reference tests/test_plan7/test_tophits.py:112-160, 293-327, 343-357, 383-450 and test_hit.py:94-150."""
    hmm = models["PF02826"][0]
    base = host_pipeline.host_search(oracle, hmm, proteome)
    hits = base.copy()
    assert hits.mode == "search" and hits.strand is None and bool(hits)
    assert hits.query is hmm and hits.query.name == hmm.name and hits.query.M == hmm.M
    with pytest.raises(IndexError):
        hits[len(hits) + 1]
    with pytest.raises(IndexError):
        hits[-len(hits) - 1]
    assert hits[len(hits) - 1].name == hits[-1].name == "938293.PRJEB85.HG003687_187"
    assert (len(hits.included), len(hits.reported)) == (15, 22)
    # sorting
    assert hits.is_sorted() and not hits.is_sorted(by="seqidx")
    hits.sort(by="seqidx")
    assert hits.is_sorted(by="seqidx") and not hits.is_sorted(by="key")
    assert [h.seqidx for h in hits] == sorted(h.seqidx for h in hits)
    hits.sort()
    assert hits.is_sorted() and [h.name for h in hits] == [h.name for h in base]
    with pytest.raises(ValueError):
        hits.sort(by="nonsense")
    with pytest.raises(ValueError):
        hits.is_sorted(by="nonsense")
    # pickling round trip
    again = pickle.loads(pickle.dumps(hits))
    assert [(h.name, h.score, h.evalue, h.included) for h in again] == [(h.name, h.score, h.evalue, h.included) for h in hits]
    # ... and a blob written by another version of the library is rejected, not mis-parsed (the layout follows the ABI's
    # configuration record: the ABI version and the record's size travel behind the magic)
    blob = bytearray(hits.to_bytes())
    assert int.from_bytes(blob[4:8], "little") == 8
    blob[4:8] = (6).to_bytes(4, "little")
    with pytest.raises(ValueError, match="another library version"):
        plan7.TopHits.from_bytes(bytes(blob), None)
    blob[0:4] = b"nope"
    with pytest.raises(ValueError):
        plan7.TopHits.from_bytes(bytes(blob), None)
    # manual flags
    hits[0].reported = False
    assert len(hits.reported) == 21 and not hits[0].reported and len(hits.included) == 14
    hits[0].reported = True
    hits[0].included = True
    assert (len(hits.included), len(hits.reported)) == (15, 22)
    hits[0].included = False
    assert (len(hits.included), len(hits.reported)) == (14, 22) and hits[0].reported
    hits[0].included = True
    hits[0].dropped = True
    assert (len(hits.included), len(hits.reported)) == (14, 22) and hits[0].dropped and not hits[0].included
    hits[0].dropped = False
    assert (len(hits.included), len(hits.reported)) == (14, 22) and not hits[0].dropped and not hits[0].included
    hits[0].included = True
    hits[0].duplicate = True
    assert (len(hits.included), len(hits.reported)) == (14, 21) and hits[0].duplicate
    hits[0].duplicate = False
    assert (len(hits.included), len(hits.reported)) == (14, 21) and not hits[0].duplicate
    hits.threshold()
    assert (len(hits.included), len(hits.reported)) == (15, 22)
    assert (len(base.included), len(base.reported)) == (15, 22)          # the copy was independent


def test_host_hit_text_setters_and_alignment_rendering(models, oracle, proteome):
    """reference tests/test_plan7/test_hit.py:29-93 and test_alignment.py:33-39."""
    hmm = models["PF02826"][0]
    base = host_pipeline.host_search(oracle, hmm, proteome)
    hits = base.copy()
    hit = hits[-1]
    assert hit.name == hits[-1].name == base[-1].name == "938293.PRJEB85.HG003687_187" and hit.length == 281
    hits[-1].name = "new name"
    assert hit.name == hits[-1].name == "new name" and base[-1].name == "938293.PRJEB85.HG003687_187"
    with pytest.raises(TypeError):
        hit.name = None
    assert hit.accession is None
    hits[-1].accession = "NEW"
    assert hit.accession == hits[-1].accession == "NEW" and base[-1].accession is None
    hits[-1].accession = None
    assert hit.accession is None
    assert hit.description.startswith("# 202177 # 203019 #")
    hits[-1].description = None
    assert hit.description is None and base[-1].description.startswith("# 202177 # 203019 #")
    hits[-1].description = "NEW"
    assert hit.description == hits[-1].description == "NEW"
    buf = io.BytesIO()
    hits.write(buf, format="targets", header=False)
    last = buf.getvalue().decode().splitlines()[-1]
    assert last.startswith("new name") and last.endswith(" NEW")
    ali = base[0].best_domain.alignment
    lines = str(ali).splitlines()
    assert len(lines) == 5                                   # CS, model, match, target, PP
    assert lines[0].rstrip().endswith("CS") and lines[4].rstrip().endswith("PP")
    assert lines[1].strip().startswith(base.query.name) and lines[3].strip().startswith(base[0].name)
    assert len({len(l) - len(l.lstrip()) + l.lstrip().find(" ") for l in (lines[1], lines[3])}) == 1   # names right-aligned


def test_host_write_rrefam_tables_per_query(models, oracle, proteome):
    """RREFam.tbl / RREFam.domtbl hold the rows of every query model (one p7_tophits_Tabular* call per query, header
    once): each model's `TopHits.write(header=False)` must reproduce its rows as text."""
    from conftest import GOLDEN
    for fmt, table, qcol in (("targets", "RREFam.tbl", 2), ("domains", "RREFam.domtbl", 3)):
        golden_rows = [l for l in open(GOLDEN / "tables" / table).read().splitlines() if l and not l.startswith("#")]
        nrows = 0
        for hmm in models["RREFam"]:
            hits = host_pipeline.host_search(oracle, hmm, proteome)
            buf = io.BytesIO()
            hits.write(buf, format=fmt, header=False)
            got = buf.getvalue().decode().splitlines()
            want = [l for l in golden_rows if l.split()[qcol] == hmm.name]
            assert got == want, hmm.name
            nrows += len(got)
        assert nrows == len(golden_rows)
