"""End-to-end parity of Pipeline.search_hmm / hmmsearch with the reference's golden tables
(real HMMER output: tests/golden/tables, reference tests/test_hmmer.py:51-238) and with the oracle's cascade."""
import itertools
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np
import pytest

from conftest import golden_table, synthetic_block
from golden_checks import check_domtbl as _check_domtbl, check_tbl as _check_tbl
from pyhmmer_amd import _lib, easel, errors, hmmer, plan7
from test_oracle_golden import STAGE_COUNTS

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["one-target-per-wavefront kernels for small blocks", "lane-per-target kernels"])
def kernel_family(request):
    """Blocks of up to 65,536 targets normally take the wave-per-target MSV / Viterbi kernels;
    the second pass sends the same cases through the lane-per-target MSV and the packed Viterbi kernels (test seam
    p7x_debug_set_option "small_block")."""
    from pyhmmer_amd import _lib
    _lib.set_debug_option("small_block", 0 if request.param.startswith("lane") else -1)
    yield
    _lib.set_debug_option("small_block", -1)


def test_pf02826_hits_and_domains_match_hmmer(models, proteome):
    hmm = models["PF02826"][0]
    hits = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, proteome)
    assert len(hits) == 22                                   # reference test_hmmer.py:114
    assert hits.Z == len(proteome) and hits.searched_residues == proteome.total_length()
    assert tuple(hits.stage_counts.values()) == STAGE_COUNTS[hmm.name]
    _check_tbl(hits, golden_table("PF02826.tbl"))
    _check_domtbl(hits, golden_table("PF02826.domtbl", kind="domtbl"))


def test_thioesterase_inline_known_answer(models, proteome):
    """reference test_hmmer.py:51-106."""
    hmm = models["Thioesterase"][0]
    hits = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, proteome)
    assert len(hits) == 1
    hit = hits[0]
    assert hit.name == "938293.PRJEB85.HG003687_113"
    assert hit.score == pytest.approx(8.6, abs=0.1) and hit.bias == pytest.approx(1.5, abs=0.1)
    assert hit.evalue == pytest.approx(0.096, abs=0.01)
    assert len(hit.domains) == 1
    d = hit.domains[0]
    assert d.score == pytest.approx(8.1, abs=0.1) and d.bias == pytest.approx(1.5, abs=0.1)
    assert d.i_evalue == pytest.approx(0.14, abs=0.005) and d.c_evalue == pytest.approx(6.5e-05, abs=0.005)
    assert (d.alignment.target_from, d.alignment.target_to, d.alignment.target_length) == (115, 129, 261)
    assert (d.alignment.hmm_from, d.alignment.hmm_to, d.alignment.hmm_length) == (79, 93, 243)
    assert (d.env_from, d.env_to) == (115, 129)
    assert d.alignment.hmm_sequence == "GWSfGGvlAyEmArq"
    assert d.alignment.identity_sequence == "G+S+GG +A ++A++"
    assert d.alignment.target_sequence == "GHSMGGSVAVAIAHE"
    assert d.alignment.posterior_probabilities == "9************96"
    tophits_T5 = plan7.Pipeline(hmm.alphabet, T=5).search_hmm(hmm, proteome)     # reference _hmmsearch.py:365-367
    assert tophits_T5[0].score == pytest.approx(8.601, abs=0.02)


def test_rrefam_hits_and_domains_match_hmmer(models, proteome):
    """reference test_hmmer.py:161-198 (bias filter with a non-zero composition vector)."""
    db = plan7.SequenceDatabase(proteome)
    pli = plan7.Pipeline(proteome.alphabet)
    all_hits = [pli.search_hmm(hmm, db) for hmm in models["RREFam"]]
    for hmm, hits in zip(models["RREFam"], all_hits):
        assert tuple(hits.stage_counts.values()) == STAGE_COUNTS[hmm.name]
        _check_tbl(hits, golden_table("RREFam.tbl", hmm.name))
        _check_domtbl(hits, golden_table("RREFam.domtbl", hmm.name, kind="domtbl"))
        for h in hits:
            assert h.hits.Z == len(proteome)


def test_cascade_decisions_match_oracle_on_synthetic_block(models, oracle):
    hmm = models["KR"][0]
    bg = plan7.Background(hmm.alphabet)
    blk = synthetic_block(30000, 300, seed=5)
    hits = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, blk)
    recs, ctr = oracle.OracleProfile(hmm, bg, 400).cascade_block(blk.packed(), want_records=False)
    assert tuple(hits.stage_counts.values()) == (ctr.n_past_msv, ctr.n_past_bias, ctr.n_past_vit, ctr.n_past_fwd)


def test_shard_merge_equals_whole(models, proteome):
    """reference tests/test_plan7/test_tophits.py:191-224: merged shards == one search, field by field."""
    hmm = models["PF02826"][0]
    pli = plan7.Pipeline(hmm.alphabet)
    whole = pli.search_hmm(hmm, proteome)
    parts = [pli.search_hmm(hmm, proteome[a:b]) for a, b in ((0, 1000), (1000, 2000), (2000, 2100))]
    merged = parts[0].merge(*parts[1:])
    assert merged.Z == whole.Z == 2100 and merged.domZ == whole.domZ
    assert merged.stage_counts == whole.stage_counts and merged.searched_residues == whole.searched_residues
    assert len(merged) == len(whole)
    for a, b in zip(merged, whole):
        assert (a.name, a.score, a.pre_score, a.sum_score, a.evalue, a.reported, a.included) == \
               (b.name, b.score, b.pre_score, b.sum_score, b.evalue, b.reported, b.included)
        for da, dbb in itertools.zip_longest(a.domains, b.domains):
            assert (da.env_from, da.env_to, da.score, da.c_evalue, da.i_evalue, da.reported, da.included) == \
                   (dbb.env_from, dbb.env_to, dbb.score, dbb.c_evalue, dbb.i_evalue, dbb.reported, dbb.included)
            assert da.alignment.target_sequence == dbb.alignment.target_sequence
    # through the public entry point, sharded "over devices" (same device listed twice)
    via = list(hmmer.hmmsearch(hmm, proteome, devices=[0, 0]))[0]
    assert [h.name for h in via] == [h.name for h in whole] and via.Z == 2100


def test_hmm_vs_optimized_profile_query(models, proteome):
    """reference test_hmmer.py:201-237: HMM query == OptimizedProfile query, exact equality."""
    hmm = models["PF02826"][0]
    bg = plan7.Background(hmm.alphabet)
    prof = plan7.Profile(hmm.M, hmm.alphabet)
    prof.configure(hmm, bg, 100)
    pli = plan7.Pipeline(hmm.alphabet)
    a = pli.search_hmm(hmm, proteome)
    b = pli.search_hmm(prof.to_optimized(), proteome)
    assert len(a) == len(b) == 22
    for x, y in zip(a, b):
        assert (x.name, x.score, x.pre_score, x.sum_score, x.evalue) == (y.name, y.score, y.pre_score, y.sum_score, y.evalue)


def test_z_and_bit_cutoffs(models, proteome):
    """reference test_pipeline.py:151-195."""
    hmm = models["PF02826"][0]
    hits = plan7.Pipeline(hmm.alphabet, Z=25).search_hmm(hmm, proteome)
    assert hits.Z == 25
    ga = plan7.Pipeline(hmm.alphabet, bit_cutoffs="gathering").search_hmm(hmm, proteome)
    assert all(h.score >= 25.1 for h in ga.reported) and len(ga.reported) == 7
    thio = models["Thioesterase"][0]
    with pytest.raises(errors.MissingCutoffs):
        plan7.Pipeline(thio.alphabet, bit_cutoffs="gathering").search_hmm(thio, proteome)
    with pytest.raises(errors.AlphabetMismatch):
        plan7.Pipeline(easel.Alphabet.dna()).search_hmm(hmm, proteome)


def test_hmmsearch_query_pipeline_matches_back_to_back_searches(models, proteome):
    """hmmsearch overlaps the device stage of the next query with the host stage of the current one; the hits must
    be those of plain sequential searches, in query order, also across a resident SequenceDatabase."""
    queries = models["RREFam"] + models["PF02826"] + models["KR"] + models["RREFam"][:3]
    seq = [h for h in hmmer.hmmsearch(queries, proteome, pipeline_depth=0)]
    for depth, target in ((2, proteome), (4, plan7.SequenceDatabase(proteome))):
        got = [h for h in hmmer.hmmsearch(iter(queries), target, pipeline_depth=depth)]
        assert len(got) == len(seq) == len(queries)
        for q, a, b in zip(queries, got, seq):
            assert a.query.name == q.name
            assert a.to_bytes()[:0] == b""                  # serialisable
            assert [(h.name, h.score, h.evalue, len(h.domains)) for h in a] == [(h.name, h.score, h.evalue, len(h.domains)) for h in b]
            assert a.stage_counts == b.stage_counts


def test_hmmsearch_pipeline_forwards_errors_and_can_be_abandoned(models, proteome):
    queries = [models["PF02826"][0], models["KR"][0], models["PF02826"][0]]     # KR.hmm has no GA cutoffs
    it = hmmer.hmmsearch(queries, proteome, bit_cutoffs="gathering")
    assert len(next(it)) > 0
    with pytest.raises(errors.MissingCutoffs):
        next(it)
    it = hmmer.hmmsearch(models["RREFam"], proteome)       # closing the generator early must not leak or hang
    next(it)
    it.close()


def test_search_with_pressed_profiles_equals_search_with_text_models(models, proteome):
    """OptimizedProfile records read from a pressed database (.h3f/.h3p) are complete queries."""
    from conftest import GOLDEN
    db = plan7.SequenceDatabase(proteome)
    pli = plan7.Pipeline(proteome.alphabet)
    for name in ("PF02826", "RREFam"):
        with plan7.HMMPressedFile(GOLDEN / "db" / f"{name}.hmm") as pressed:
            for om, hmm in zip(pressed, models[name]):
                a, b = pli.search_hmm(om, db), pli.search_hmm(hmm, db)
                assert a.stage_counts == b.stage_counts
                assert [(h.name, h.score, h.evalue, [(d.env_from, d.env_to, d.score, d.alignment.hmm_sequence) for d in h.domains]) for h in a] == \
                       [(h.name, h.score, h.evalue, [(d.env_from, d.env_to, d.score, d.alignment.hmm_sequence) for d in h.domains]) for h in b]


def test_hmmscan_rrefam_matches_hmmer_scan_table(models, proteome):
    """reference test_hmmer.py:836-904 (hmmscan of the RREFam profiles) against tables/RREFam.scan.tbl, which real
    HMMER produced from the fixture proteome: per query sequence the same models, in the same order, with the table's
    score / bias / E-value (Z = number of profiles)."""
    from conftest import GOLDEN
    import io
    text_rows = [l for l in open(GOLDEN / "tables" / "RREFam.scan.tbl").read().splitlines() if l and not l.startswith("#")]
    expected = {}
    for row in golden_table("RREFam.scan.tbl"):
        expected.setdefault(row[2], []).append(row)
    nq = 0
    for source in (models["RREFam"], plan7.HMMPressedFile(GOLDEN / "db" / "RREFam.hmm")):
        seen = set()
        for seq, hits in zip(proteome, hmmer.hmmscan(proteome, source)):
            assert hits.query is seq and hits.Z == 10
            rows = expected.get(seq.name, [])
            assert [h.name for h in hits.reported] == [r[0] for r in rows], seq.name
            for h, r in zip(hits.reported, rows):
                assert h.score == pytest.approx(float(r[5]), abs=0.1) and h.bias == pytest.approx(float(r[6]), abs=0.1)
                assert h.evalue == pytest.approx(float(r[4]), rel=0.06)
                assert h.accession == (None if r[1] == "-" else r[1])
                assert len(h.domains) == int(r[15]) and h.domains[0].alignment.hmm_name == h.name
                assert h.domains[0].alignment.target_name == seq.name
            if rows:
                seen.add(seq.name)
                # the --tblout text itself (one p7_tophits_TabularTargets call per query sequence, header once)
                out = io.BytesIO()
                hits.write(out, format="targets", header=False)
                want = [l for l in text_rows if l.split()[2] == seq.name]
                assert out.getvalue().decode().splitlines() == want
            nq += 1
        assert seen == set(expected)
    assert nq == 2 * len(proteome)


def test_hmmscan_equals_transposed_hmmsearch(models, proteome):
    """Per (model, sequence) pair scan and search run the same comparison: scores and domains agree, only the roles of
    query and target (names, Z, E-values, accounting) differ.  Also scan_seq for a single query."""
    sub = proteome
    profs = models["RREFam"] + models["PF02826"]
    by_pair = {}
    for hmm, hits in zip(profs, hmmer.hmmsearch(profs, sub, E=1e9, domE=1e9)):
        for h in hits:
            by_pair[(hmm.name, h.name)] = (h.score, h.bias, [(d.env_from, d.env_to, d.score) for d in h.domains])
    got = {}
    all_hits = list(hmmer.hmmscan(sub, profs, E=1e9, domE=1e9))
    for seq, hits in zip(sub, all_hits):
        assert hits.searched_models == len(profs) and hits.searched_sequences == 1 and hits.searched_residues == len(seq)
        assert hits.searched_nodes == sum(p.M for p in profs)
        for h in hits:
            got[(h.name, seq.name)] = (h.score, h.bias, [(d.env_from, d.env_to, d.score) for d in h.domains])
    assert got == by_pair and len(got) > 10
    q = next(s for s, hits in zip(sub, all_hits) if len(hits))
    one = plan7.Pipeline(sub.alphabet, E=1e9, domE=1e9).scan_seq(q, profs)
    ref = next(hits for s, hits in zip(sub, all_hits) if s is q)
    assert [(h.name, h.score) for h in one] == [(h.name, h.score) for h in ref]


def test_concurrent_feeders_and_finishers_keep_order_and_forward_errors(models, proteome):
    """Several searches in flight on separate streams (the hmmscan configuration) must return the results of the
    sequential loop, in order; an error of one query surfaces at its position; abandoning the generator is clean."""
    queries = (models["RREFam"] + models["PF02826"] + models["KR"]) * 3
    want = [[(h.name, h.score) for h in hits] for hits in hmmer.hmmsearch(queries, proteome, pipeline_depth=0)]
    got = [[(h.name, h.score) for h in hits] for hits in hmmer.hmmsearch(iter(queries), proteome, pipeline_depth=12, feeders=6)]
    assert got == want
    bad = [models["PF02826"][0]] * 5 + [models["KR"][0]] + [models["PF02826"][0]] * 5       # KR.hmm has no GA cutoffs
    it = hmmer.hmmsearch(bad, proteome, bit_cutoffs="gathering", pipeline_depth=8, feeders=4)
    for _ in range(5):
        assert len(next(it)) > 0
    with pytest.raises(errors.MissingCutoffs):
        next(it)
    it = hmmer.hmmsearch(queries, proteome, pipeline_depth=8, feeders=4)
    next(it); next(it)
    it.close()


def test_hmmscan_with_gathering_cutoffs(models, proteome):
    """Bit-score cutoffs are model specific: in scan orientation they travel with each profile's result."""
    hmm = models["PF02826"][0]
    search = hmmer.hmmsearch([hmm], proteome, bit_cutoffs="gathering")
    want = sorted(h.name for h in next(search).reported)
    assert len(want) == 7
    got = sorted(seq.name for seq, hits in zip(proteome, hmmer.hmmscan(proteome, [hmm], bit_cutoffs="gathering"))
                 if [h for h in hits.reported if h.name == hmm.name])
    assert got == want
    with pytest.raises(errors.MissingCutoffs):
        list(hmmer.hmmscan(proteome, [hmm, models["KR"][0]], bit_cutoffs="gathering"))


@pytest.mark.parametrize("fmt,table", [("targets", "PF02826.tbl"), ("domains", "PF02826.domtbl")])
def test_written_tables_equal_hmmer_output_text(models, proteome, fmt, table):
    """reference tests/test_plan7/test_tophits.py:359-381, through the device path: every line as text."""
    import io
    from conftest import GOLDEN
    hmm = models["PF02826"][0]
    hits = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, proteome)
    buf = io.BytesIO()
    hits.write(buf, format=fmt)
    got = buf.getvalue().decode().splitlines()
    want = open(GOLDEN / "tables" / table).read().splitlines()
    while want[-1].startswith("#"):
        want.pop()
    assert got == want
    # scan mode: the model names are the "targets" and the sequence is the query (RREFam.scan.tbl layout)
    seq = next(s for s in proteome if s.name == "938293.PRJEB85.HG003691_78")
    scan = list(hmmer.hmmscan([seq], models["RREFam"]))[0]
    assert scan.mode == "scan"
    out = io.BytesIO()
    scan.write(out, format="targets", header=False)
    rows = [l.split() for l in out.getvalue().decode().splitlines()]
    want_rows = [r for r in golden_table("RREFam.scan.tbl") if r[2] == seq.name]
    assert [(r[0], r[1], r[2]) for r in rows] == [(r[0], r[1], r[2]) for r in want_rows]


def test_one_thread_keeps_several_searches_in_flight(models, proteome):
    """p7x_search_block_enqueue / _wait: stage 1 of every RREFam model is queued by this thread before it waits for any
    of them; finished out of order; the results equal the blocking searches.  An un-waited handle can be dropped."""
    db = plan7.SequenceDatabase(proteome)
    pli = plan7.Pipeline(proteome.alphabet)
    want = [pli.search_hmm(hmm, db) for hmm in models["RREFam"]]
    pend = [pli._search_enqueue(hmm, db) for hmm in models["RREFam"]]
    got = [None] * len(pend)
    for i in reversed(range(len(pend))):
        if i % 2:
            plan7.Pipeline._search_wait(pend[i])
            plan7.Pipeline._search_wait(pend[i])          # idempotent
        got[i] = plan7.Pipeline._search_finish(pend[i])   # finish waits when the caller did not
    for a, b in zip(got, want):
        assert [(h.name, h.score, h.evalue, len(h.domains)) for h in a] == [(h.name, h.score, h.evalue, len(h.domains)) for h in b]
        assert a.stage_counts == b.stage_counts
    from pyhmmer_amd import _lib
    dropped = pli._search_enqueue(models["RREFam"][0], db)
    _lib.lib().p7x_pending_destroy(dropped[0])
    again = pli.search_hmm(models["RREFam"][0], db)
    assert [h.name for h in again] == [h.name for h in want[0]]
    # hmmscan with a window larger than the number of models, and with one model in flight
    seqs = [s for s in proteome][:64]
    ref = [[(h.name, round(h.score, 3)) for h in th] for th in hmmer.hmmscan(seqs, models["RREFam"], feeders=1, pipeline_depth=1, window=1)]
    for feeders, depth, window in ((1, 64, 64), (3, 5, 2), (2, 2, 8)):
        res = [[(h.name, round(h.score, 3)) for h in th] for th in hmmer.hmmscan(seqs, models["RREFam"], feeders=feeders, pipeline_depth=depth, window=window)]
        assert res == ref


def _hit_fields(hits):
    return [(h.name, h.score, h.pre_score, h.sum_score, h.evalue, h.reported, h.included, len(h.domains),
             [(d.env_from, d.env_to, d.score, d.alignment.hmm_from, d.alignment.hmm_to, d.alignment.target_from, d.alignment.target_to)
              for d in h.domains]) for h in hits]


def test_batched_search_equals_single_searches(models, proteome):
    """p7x_search_batch_enqueue / _finish: 16 fixture models of 27 ... 430 nodes (several instantiations of every
    kernel family in one launch set, lanes sorted by length inside the library) give the hit lists, domains, stage
    counts and accounting of 16 separate searches (both kernel families: the autouse fixture above)."""
    db = plan7.SequenceDatabase(proteome)
    pli = plan7.Pipeline(proteome.alphabet)
    qs = models["KR"] + models["RREFam"] + models["PF02826"] + models["Thioesterase"] + models["LuxC"] + models["RREFam"][:2]
    want = [pli.search_hmm(q, db) for q in qs]
    pend = pli._search_enqueue_batch(qs, db)
    plan7.Pipeline._search_wait(pend)
    got = plan7.Pipeline._search_finish_batch(pend)
    assert len(got) == len(qs)
    for q, a, b in zip(qs, got, want):
        assert a.query is q
        assert a.stage_counts == b.stage_counts, q.name
        assert (a.Z, a.domZ, a.searched_sequences, a.searched_residues, a.searched_nodes) == \
               (b.Z, b.domZ, b.searched_sequences, b.searched_residues, b.searched_nodes)
        assert _hit_fields(a) == _hit_fields(b), q.name
    assert sum(len(h) for h in got) > 30
    # through the public entry point, every batch size from "one at a time" to "all in one"
    for batch in (1, 3, 64):
        res = list(hmmer.hmmsearch(qs, db, batch=batch))
        assert [_hit_fields(a) for a in res] == [_hit_fields(b) for b in want], batch
    # an un-waited batch can be dropped; a failing member (no gathering cutoffs) fails the batch before anything is queued
    from pyhmmer_amd import _lib
    _lib.lib().p7x_pending_destroy(pli._search_enqueue_batch(qs[:3], db)[0])
    with pytest.raises(errors.MissingCutoffs):
        plan7.Pipeline(proteome.alphabet, bit_cutoffs="gathering")._search_enqueue_batch(qs, db)


def test_batch_with_repeated_and_fresh_profile_objects(models, proteome):
    """The device images of a batch's new profiles are laid out together, share one slab and go up in one copy
    (get_dev_profiles).  The same OptimizedProfile object several times in one batch gets one image; dropping most of
    the profiles of a batch leaves the survivor's image intact, also after the pool has handed memory to new images."""
    import gc
    bg = plan7.Background(proteome.alphabet)
    db = plan7.SequenceDatabase(proteome)
    pli = plan7.Pipeline(proteome.alphabet)
    base = models["PF02826"] + models["RREFam"][:6] + models["KR"] + models["LuxC"]
    oms = [plan7.OptimizedProfile(h, bg, 400) for h in base]                  # no device image yet
    qs = oms + [oms[0], oms[3], oms[0]]                                       # the same objects again inside the batch
    got = plan7.Pipeline._search_finish_batch(pli._search_enqueue_batch(qs, db))
    want = [pli.search_hmm(h, db) for h in base]
    want += [want[0], want[3], want[0]]
    assert [_hit_fields(a) for a in got] == [_hit_fields(b) for b in want]
    assert sum(len(a) for a in got) > 20
    keep, ref = oms[1], _hit_fields(want[1])
    del oms, qs, got
    gc.collect()
    for _ in range(3):                                                        # new batches: their slabs come from the pool
        fresh = [plan7.OptimizedProfile(h, bg, 400) for h in base]
        again = plan7.Pipeline._search_finish_batch(pli._search_enqueue_batch(fresh, db))
        assert [_hit_fields(a) for a in again] == [_hit_fields(b) for b in want[:len(base)]]
        del fresh, again
        gc.collect()
        assert _hit_fields(pli.search_hmm(keep, db)) == ref


def test_batched_search_more_survivors_than_the_shared_buffers(models, proteome):
    """With the filters switched off (F1 = F2 = F3 = 1) every target of every lane reaches Backward: the lanes' rows
    do not fit the shared arena and the per-lane retry path has to produce the same lists as separate searches."""
    db = plan7.SequenceDatabase(proteome)
    pli = plan7.Pipeline(proteome.alphabet, F1=1.0, F2=1.0, F3=1.0, bias_filter=False)
    qs = models["RREFam"][:3] + models["PF02826"]
    want = [pli.search_hmm(q, db) for q in qs]
    pend = pli._search_enqueue_batch(qs, db)
    got = plan7.Pipeline._search_finish_batch(pend)           # finish waits when the caller did not
    for q, a, b in zip(qs, got, want):
        assert a.stage_counts == b.stage_counts and a.stage_counts["fwd"] == len(proteome), q.name
        assert _hit_fields(a) == _hit_fields(b), q.name


def test_optimized_profile_block_scan(models, proteome):
    """reference tests/test_plan7/test_pipeline.py:229-240 (scan_seq over an OptimizedProfileBlock) and the container
    protocol of plan7.pyx:5121-5338."""
    bg = plan7.Background(proteome.alphabet)
    oms = [plan7.OptimizedProfile(h, bg) for h in models["RREFam"]]
    block = plan7.OptimizedProfileBlock(proteome.alphabet, oms)
    assert len(block) == 10 and block[3] is oms[3] and oms[4] in block and block.index(oms[4]) == 4
    assert len(block[2:5]) == 3 and block.copy() == block
    block.append(block.pop())
    with pytest.raises(TypeError):
        block.append(models["RREFam"][0])
    with pytest.raises(errors.AlphabetMismatch):
        plan7.OptimizedProfileBlock(easel.Alphabet.dna(), oms)
    seq = next(s for s in proteome if s.name == "938293.PRJEB85.HG003691_78")
    pli = plan7.Pipeline(proteome.alphabet)
    hits = pli.scan_seq(seq, block)
    ref = list(hmmer.hmmscan([seq], models["RREFam"], batch=1))[0]
    assert len(hits) >= 1 and _hit_fields(hits) == _hit_fields(ref) and hits.mode == "scan"


@pytest.mark.timeout(600)
def test_config3_profile_library_against_a_proteome(proteome, request):
    """BASELINE configs[2] at full profile count (SURVEY.md 8d "config 3"): a 20,000-entry profile library (the
    synthetic, device-calibrated Pfam stand-in of bench_workloads.py; M ~ lognormal, 20 ... 2000 nodes) scanned
    against the 2,100-sequence fixture proteome through hmmer.hmmscan in batches of 256.  Properties that do not
    depend on the size: every per-sequence hit list carries Z = number of profiles; the per-profile stage counts and
    hits of the batched run equal those of one-profile-at-a-time searches (sampled entries across the length range);
    the hits of a sequence across all profiles equal the transposed per-profile hits."""
    import bench_workloads as bw
    if "lane-per-target" in request.node.name:
        pytest.skip("one pass is enough: a 20,000-profile batch run picks its kernels itself")
    n = 20000
    hmms, lengths, templates = bw.make_library(n, count=n)
    assert len(hmms) == n and int(lengths.min()) >= 20 and int(lengths.max()) > 1000
    bg = plan7.Background(proteome.alphabet)
    block = plan7.OptimizedProfileBlock(proteome.alphabet, (plan7.OptimizedProfile(h, bg, 400) for h in hmms))
    db = plan7.SequenceDatabase(proteome)
    # search orientation over the whole library, batched: stage counts per profile
    batched = list(hmmer.hmmsearch(block, db, batch=256, pipeline_depth=4, feeders=2))
    assert len(batched) == n
    # ... and with the library's own choices (round 6: the first batch decides the feeders -- three here --, a sized source of
    # 1,000 queries or more fills its batches to 64 profiles): the same results, query by query
    default = list(hmmer.hmmsearch(block, db))
    assert hmmer.pipeline_stats()["feeders"] == 3
    assert [h.stage_counts for h in default] == [h.stage_counts for h in batched]
    assert all(_hit_fields(a) == _hit_fields(b) for a, b in zip(default, batched))
    sample = sorted(set(list(range(0, n, 997)) + [int(np.argmax(lengths)), int(np.argmin(lengths))]))
    pli = plan7.Pipeline(proteome.alphabet)
    for e in sample:
        single = pli.search_hmm(block[e], db)
        assert batched[e].stage_counts == single.stage_counts, (e, hmms[e].M)
        assert _hit_fields(batched[e]) == _hit_fields(single), (e, hmms[e].M)
    total = {k: sum(h.stage_counts[k] for h in batched) for k in ("msv", "bias", "vit", "fwd")}
    assert total["msv"] >= total["bias"] >= total["vit"] >= total["fwd"] > 0
    # the MSV filter passes about F1 of the comparisons for calibrated models (0.02; composition and length effects allowed for)
    assert 0.005 < total["msv"] / (n * len(proteome)) < 0.06
    # scan orientation: per-sequence lists, Z = number of profiles, same (profile, sequence) pairs
    scanned = list(hmmer.hmmscan(proteome, block))          # default: as many profiles per batch as the workspace allows (4,096 here)
    assert len(scanned) == len(proteome) and all(h.Z == n for h in scanned[:50])
    pairs_scan = {(hit.name, q.name) for q, th in zip(proteome, scanned) for hit in th}
    pairs_search = {(hmms[e].name, hit.name) for e, th in enumerate(batched) for hit in th}
    # reportability differs (scan: E-values with Z = 20000 profiles; search: Z = 2100 sequences): compare the strong pairs
    strong_scan = {(hit.name, q.name) for q, th in zip(proteome, scanned) for hit in th if hit.score > 40}
    strong_search = {(hmms[e].name, hit.name) for e, th in enumerate(batched) for hit in th if hit.score > 40}
    assert strong_scan == strong_search and len(pairs_scan) > 0 and len(pairs_search) > 0


def test_forward_threshold_ties_are_decided_in_the_reference_order(models, oracle, proteome):
    """F3 placed exactly on the (reference-order) Forward P-value of a target, and one ulp below it: the number of targets
    past the Forward filter is what the oracle's arithmetic gives, although the device sums Forward in another order
    (cfg.f3_guard: the device passes the band around F3, the host stage re-scores it)."""
    import math
    hmm = models["PF02826"][0]
    bg = plan7.Background(hmm.alphabet)
    op = oracle.OracleProfile(hmm, bg, 400)
    recs, _ = op.cascade_block(proteome.packed(), F3=1.0)
    ftau, flam = float(hmm.evalue_parameters.f_tau), float(hmm.evalue_parameters.f_lambda)

    def pval(rec):
        sc = np.float32((np.float64(np.float32(rec.fwdsc) - np.float32(rec.filtersc))) / 0.69314718055994529)
        return math.exp(-flam * (float(sc) - ftau)) if float(sc) >= ftau else 1.0

    cand = sorted((t for t in range(len(proteome)) if recs[t].stage >= 4 and 1e-9 < pval(recs[t]) < 1e-2), key=lambda t: pval(recs[t]))
    assert len(cand) >= 4
    for t in cand[:: max(1, len(cand) // 4)]:
        P = pval(recs[t])
        for F3 in (P, float(np.nextafter(P, 0.0))):
            want = sum(1 for u in range(len(proteome)) if recs[u].stage >= 4 and not (pval(recs[u]) > F3))
            hits = plan7.Pipeline(hmm.alphabet, F3=F3).search_hmm(hmm, proteome)
            assert hits.stage_counts["fwd"] == want, (proteome[t].name, F3)


def test_hmmscan_deals_profile_batches_over_devices(models, proteome):
    """Multi-device scan orientation (few sequences, many profiles -> shard the profiles): every device holds
    the whole query block and takes batches of profiles in turn.  With the one device of the test box listed twice the
    dealing, the ordering and the transposition are exercised; results must equal the single-device scan."""
    profs = (models["RREFam"] + models["PF02826"] + models["Thioesterase"]) * 3
    sub = proteome[:600]
    one = list(hmmer.hmmscan(sub, profs, batch=4))
    two = list(hmmer.hmmscan(sub, profs, devices=[0, 0], batch=4))
    assert len(one) == len(two) == len(sub)
    for a, b in zip(one, two):
        assert a.searched_models == b.searched_models == len(profs)
        assert [(h.name, h.score, h.evalue, len(h.domains)) for h in a] == [(h.name, h.score, h.evalue, len(h.domains)) for h in b]
    db = hmmer.ReplicatedDatabase(sub, [0, 0])
    assert [sh.device for sh in db.shards] == [0, 0] and all(len(sh.block) == len(sub) for sh in db.shards)


def test_clustered_envelopes_on_the_device_give_the_same_tables(models, proteome):
    """With the stochastic ensembles on the host workers (host_ensembles) their clustered envelopes can be rescored by the
    envelope kernel in a second round or by the host workers (option "device_clustered"); with the ensembles on the device
    (the default) the second round always runs there.  Hits, domains and the written tables are the same every way (the
    golden tables contain nine such hits)."""
    import io
    from pyhmmer_amd import _lib
    hmm = models["PF02826"][0]
    try:
        _lib.set_debug_option("device_clustered", 0)
        base = plan7.Pipeline(hmm.alphabet, host_ensembles=True).search_hmm(hmm, proteome)
        _lib.set_debug_option("device_clustered", 1)
        dev = plan7.Pipeline(hmm.alphabet, host_ensembles=True).search_hmm(hmm, proteome)
    finally:
        _lib.set_debug_option("device_clustered", -1)
    alldev = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, proteome)          # ensembles and clustered envelopes on the device
    assert sum(h.nclustered for h in dev) == sum(h.nclustered for h in base) == sum(h.nclustered for h in alldev) > 0
    for res in (dev, alldev):
        _check_tbl(res, golden_table("PF02826.tbl"))
        _check_domtbl(res, golden_table("PF02826.domtbl", kind="domtbl"))
        assert [(h.name, len(h.domains), h.nenvelopes, h.noverlaps) for h in res] == [(h.name, len(h.domains), h.nenvelopes, h.noverlaps) for h in base]
        for fmt in ("targets", "domains"):
            a, b = io.BytesIO(), io.BytesIO()
            base.write(a, format=fmt); res.write(b, format=fmt)
            assert a.getvalue() == b.getvalue()
    few = plan7.Pipeline(hmm.alphabet, host_threads=4, host_ensembles=True).search_hmm(hmm, proteome)        # few host threads: the device's turn
    a, b = io.BytesIO(), io.BytesIO()
    base.write(a, format="domains"); few.write(b, format="domains")
    assert a.getvalue() == b.getvalue()


def test_target_file_is_searched_in_chunks(models, proteome, golden):
    """hmmsearch(queries, SequenceFile): the file is walked in chunks (here 100 kB of text: ten chunks for the fixture
    proteome), one chunk resident at a time; the merged hit lists equal the search of the whole block -- scores, E-values
    with the global Z, flags, domains -- and reproduce the golden table (reference plan7.pyx:6244-6252, :6456;
    test_hmmer.py:109-198 runs the same comparison in "file" mode)."""
    queries = [models["PF02826"][0], models["Thioesterase"][0]] + models["RREFam"][:3]
    whole = list(hmmer.hmmsearch(queries, proteome))
    with easel.SequenceFile(golden / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=proteome.alphabet) as sf:
        seen = []
        chunked = list(hmmer.hmmsearch(queries, sf, chunk_bytes=100_000, callback=lambda q, n: seen.append(q.name)))
        one_chunk = list(hmmer.hmmsearch(queries, sf))
    assert seen == [q.name for q in queries]
    for a, b, c in zip(chunked, whole, one_chunk):
        assert a.Z == b.Z == c.Z == 2100 and a.domZ == b.domZ
        assert a.stage_counts == b.stage_counts and a.searched_residues == b.searched_residues == 682583
        for x in (a, c):
            assert [(h.name, h.score, h.evalue, h.reported, h.included) for h in x.reported] == \
                   [(h.name, h.score, h.evalue, h.reported, h.included) for h in b.reported]
            assert [[(d.env_from, d.env_to, d.score, d.i_evalue, d.alignment.target_sequence) for d in h.domains] for h in x.reported] == \
                   [[(d.env_from, d.env_to, d.score, d.i_evalue, d.alignment.target_sequence) for d in h.domains] for h in b.reported]
    _check_tbl(chunked[0], golden_table("PF02826.tbl"))
    # an empty target file still answers every query
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".faa") as tmp:
        with easel.SequenceFile(tmp.name, digital=True, alphabet=proteome.alphabet) as sf:
            empty = list(hmmer.hmmsearch(queries[:2], sf))
    assert [len(h) for h in empty] == [0, 0] and empty[0].Z == 0


def test_hmmscan_streams_a_query_file_in_blocks(models, golden):
    """A query FILE is walked a block at a time (the reference takes its queries one at a time, plan7.pyx:6680-6737): every
    block is scanned against the whole profile database and its results are out before the next block is read; the
    per-sequence hit lists are those of a scan of the whole file as one block, in file order, whatever the block size --
    from a profile file that is rewound per block, and from a one-shot iterator of profiles."""
    import io
    path = golden / "seqs" / "938293.PRJEB85.HG003687.faa"
    abc = easel.Alphabet.amino()

    def table(results):
        out = []
        for hits in results:
            b = io.BytesIO()
            hits.write(b, format="targets", header=False)
            out.append(b.getvalue())
        return out

    with easel.SequenceFile(path, digital=True, alphabet=abc) as sf:
        whole = sf.read_block()
    profs = models["RREFam"] + models["PF02826"]
    want = table(hmmer.hmmscan(whole, profs))
    assert sum(len(t) > 0 for t in want) >= 20
    for per_block in (500, 64):
        with easel.SequenceFile(path, digital=True, alphabet=abc) as sf:
            assert table(hmmer.hmmscan(sf, iter(profs), query_block_sequences=per_block)) == want
    with easel.SequenceFile(path, digital=True, alphabet=abc) as sf, plan7.HMMFile(golden / "hmms" / "RREFam.hmm") as hf:
        got = table(hmmer.hmmscan(sf, hf, query_block_residues=200_000))           # native chunk reads, the profile file rewound per block
    assert got == table(hmmer.hmmscan(whole, models["RREFam"]))


def test_in_process_search_over_four_devices_equals_one(models, proteome):
    """hmmer.hmmsearch / hmmscan default to every visible device, one process driving all of them (round 5; the reference's
    default is the whole machine, _hmmsearch.py:384).  The one device of the test box listed four times exercises the in-process
    N-device path -- four shards enqueued by one feeder, their host stages finished side by side on the shard pool, the per-query
    merge -- for both orientations; merged == whole."""
    assert hmmer.default_devices() == list(range(max(1, _lib.lib().p7x_device_count())))
    queries = models["PF02826"] + models["RREFam"][:4] + models["Thioesterase"]
    whole = list(hmmer.hmmsearch(queries, proteome, devices=[0]))
    four = list(hmmer.hmmsearch(queries, proteome, devices=[0, 0, 0, 0]))
    assert len(whole) == len(four) == len(queries)
    for a, b in zip(four, whole):
        assert a.Z == b.Z == len(proteome) and a.domZ == b.domZ and a.stage_counts == b.stage_counts
        assert [(h.name, h.score, h.evalue, h.reported, h.included, [(d.env_from, d.env_to, d.score, d.alignment.target_sequence) for d in h.domains]) for h in a] == \
               [(h.name, h.score, h.evalue, h.reported, h.included, [(d.env_from, d.env_to, d.score, d.alignment.target_sequence) for d in h.domains]) for h in b]
    # resident shards handed in by the caller (what bench.py --inproc-devices does)
    shards = hmmer.ShardedDatabase(proteome, [0, 0, 0])
    again = list(hmmer.hmmsearch(queries, shards))
    assert [[h.name for h in t] for t in again] == [[h.name for h in t] for t in whole]
    sub = proteome[:500]
    profs = (models["RREFam"] + models["PF02826"]) * 2
    one = list(hmmer.hmmscan(sub, profs, devices=[0], batch=3))
    many = list(hmmer.hmmscan(sub, profs, devices=[0, 0, 0, 0], batch=3))
    assert [[(h.name, h.score, h.evalue) for h in t] for t in one] == [[(h.name, h.score, h.evalue) for h in t] for t in many]


def test_in_process_search_over_eight_devices_equals_one(models, proteome):
    """SURVEY.md 8(e), the node the metric is quoted on: eight devices driven by ONE process (the API's default path when eight
    are visible).  The one device of the test box listed eight times, all three orientations: hmmsearch (targets sharded by
    residues, per-query merge), hmmscan (profiles dealt), nhmmer ((target, block, strand) units dealt); merged == whole."""
    queries = models["PF02826"] + models["RREFam"][:3] + models["KR"]
    eight = [0] * 8
    whole = list(hmmer.hmmsearch(queries, proteome, devices=[0]))
    many = list(hmmer.hmmsearch(queries, proteome, devices=eight))
    for a, b in zip(many, whole):
        assert a.Z == b.Z == len(proteome) and a.domZ == b.domZ and a.stage_counts == b.stage_counts
        assert [(h.name, h.score, h.evalue, h.reported, h.included, [(d.env_from, d.env_to, d.score, d.alignment.target_sequence) for d in h.domains]) for h in a] == \
               [(h.name, h.score, h.evalue, h.reported, h.included, [(d.env_from, d.env_to, d.score, d.alignment.target_sequence) for d in h.domains]) for h in b]
    sub = proteome[:400]
    profs = (models["RREFam"] + models["PF02826"]) * 2
    one = list(hmmer.hmmscan(sub, profs, devices=[0], batch=3))
    sc8 = list(hmmer.hmmscan(sub, profs, devices=eight, batch=3))
    assert [[(h.name, h.score, h.evalue) for h in t] for t in one] == [[(h.name, h.score, h.evalue) for h in t] for t in sc8]
    import bench_workloads as bw
    from conftest import load_hmms
    dna = load_hmms("bmyD")[0]
    chrom = bw.make_chromosome(dna, 3_000_000, planted=30, seed=19)
    blk = easel.DigitalSequenceBlock(dna.alphabet, [easel.DigitalSequence(dna.alphabet, name="chr8", sequence=chrom)])
    rows = lambda hits: [(h.name, h.score, h.evalue, [(d.env_from, d.env_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]) for h in hits]
    n1 = next(hmmer.nhmmer(dna, blk, devices=[0], host_envelopes=1))
    n8 = next(hmmer.nhmmer(dna, blk, devices=eight, host_envelopes=1))
    assert len(n1) >= 20 and rows(n8) == rows(n1) and n8.stage_counts == n1.stage_counts


def test_a_failing_ensemble_workspace_sends_the_regions_to_the_host(models, proteome):
    """an allocation or launch failure of the device ensembles must not fail the search -- every region can be
    sampled by the host workers, with the same result (seam ens_fail makes DeviceEnsembleRunner::begin fail)."""
    hmm = models["PF02826"][0]
    want = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, proteome)
    assert sum(h.nclustered for h in want) > 0           # the fixture has multi-domain regions: the ensembles do run
    _lib.set_debug_option("ens_fail", 1)
    try:
        got = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, proteome)
    finally:
        _lib.set_debug_option("ens_fail", -1)
    assert [(h.name, h.score, h.nregions, h.nclustered, h.nenvelopes, [(d.env_from, d.env_to, d.score) for d in h.domains]) for h in got] == \
           [(h.name, h.score, h.nregions, h.nclustered, h.nenvelopes, [(d.env_from, d.env_to, d.score) for d in h.domains]) for h in want]


def test_the_shape_of_a_multi_class_batch_does_not_change_its_results(models, proteome):
    """One launch per tier of MSV register tiles or one per tile (msv_tiers), the batch stage by stage or class by class
    (stage_merge), the survivors' results gathered behind the cascade, by the collect half, or by both because the batch
    has more survivors than the first made room for (early_pack = 0 / 3): a batch of fourteen profiles of seven lengths --
    several kernel classes -- gives the same hit lists, alignments and accounting every way, in both orientations."""
    queries = models["RREFam"] + models["PF02826"] + models["Thioesterase"] + models["KR"] + models["LuxC"]
    assert len({q.M for q in queries}) >= 7

    def search():
        res = list(hmmer.hmmsearch(queries, proteome, devices=[0], batch=len(queries)))
        return [(t.stage_counts, t.Z, [(h.name, h.score, h.evalue, h.nregions, h.nclustered,
                                       [(d.env_from, d.env_to, d.score, d.alignment.hmm_from, d.alignment.target_sequence) for d in h.domains]) for h in t])
                for t in res]

    def scan():
        return [[(h.name, h.score, h.evalue, len(h.domains)) for h in t] for t in hmmer.hmmscan(proteome[:700], queries, devices=[0])]

    want, want_scan = search(), scan()
    assert sum(len(t[2]) for t in want) > 10 and any(want_scan)
    for name, value in (("msv_tiers", 0), ("stage_merge", 0), ("stage_merge", 1), ("early_pack", 0), ("early_pack", 3)):
        _lib.set_debug_option(name, value)
        try:
            assert search() == want, (name, value)
            assert scan() == want_scan, (name, value)
        finally:
            _lib.set_debug_option(name, -1)
