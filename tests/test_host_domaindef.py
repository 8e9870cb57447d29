"""The product's HOST half (p7_domaindef + p7_tophits restatement in libp7x) against real HMMER output.  No GPU:
the device half is stood in for by the oracle (tests/host_pipeline.py) and the rows go through the exported
C-ABI entry point p7x_postprocess_targets."""
import itertools

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
