"""oracle/p7_oracle_dd.c -- a second, independently structured implementation of domain definition (region scan, envelope
Forward / Backward, decoding, null2 by expectation, optimal-accuracy alignment, domain score; for regions that hold several
domains the ensemble of sampled tracebacks, null2 by trace and the clustering) in plain scalar C -- pinned by the
reference's own domain tables, and the product's host stage (p7x_postprocess_targets: the host twin that the device results
are compared with on the GPU) against it on synthetic targets.  No GPU."""
import numpy as np
import pytest

import host_pipeline
from conftest import golden_table, load_hmms, random_hmm
from pyhmmer_amd import easel, plan7


# Host stage and oracle both form every order-sensitive float sum in upstream's striped order (impl_sse): the same operations on
# the same operands, so scores agree to the last bit of single precision and no decision can fall differently.  What is left is
# the conversion to bits on the Python side.
TOL_BITS = 2e-5


def _golden_rows_against_the_oracle(oracle, hmm, rows, block):
    """Every row of the table: envelope, alignment and model coordinates exactly, domain score and bias at the table's print
    precision.  Returns (rows in regions that hold one domain, rows in ensemble regions)."""
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    by_name = {s.name: s for s in block}
    per_target = {}
    for r in rows:
        per_target.setdefault(r[0], []).append(r)
    single = clustered = 0
    for name, rs in per_target.items():
        envs, counts = oracle.domains(op, np.asarray(by_name[name].sequence, dtype=np.uint8))
        got = {(int(e[0]), int(e[1])): e for e in envs}
        for r in rs:
            e = got.get((int(r[19]), int(r[20])))
            assert e is not None, (name, r[19], r[20], sorted(got))
            assert (int(e[2]), int(e[3]), int(e[4]), int(e[5])) == (int(r[17]), int(r[18]), int(r[15]), int(r[16])), (name, r[15:21])
            assert abs(e[9] - float(r[13])) <= 0.051 and abs(e[10] - float(r[14])) <= 0.051, (name, r[13], r[14], e[9], e[10])
            if e[12] == 0:
                single += 1
            else:
                clustered += 1
    return single, clustered


def test_oracle_domains_reproduce_the_reference_domain_tables(oracle, proteome):
    """PF02826.domtbl and RREFam.domtbl (real hmmsearch output, reference tests/data/tables): 38 + 15 domain rows, 12 of them
    envelopes that came out of the clustering of a traceback ensemble."""
    hmm = load_hmms("PF02826")[0]
    assert _golden_rows_against_the_oracle(oracle, hmm, golden_table("PF02826.domtbl", kind="domtbl"), proteome) == (30, 8)
    single = clustered = 0
    for hmm in load_hmms("RREFam"):
        rows = golden_table("RREFam.domtbl", hmm.name, kind="domtbl")
        if rows:
            a, b = _golden_rows_against_the_oracle(oracle, hmm, rows, proteome)
            single += a; clustered += b
    assert single + clustered == 15 and clustered >= 1


def _homolog_block(hmm, nbg, nhom, seed):
    """Background sequences and sequences with one to three fragments of sampled model passes between random flanks."""
    import bench
    abc = hmm.alphabet
    rng = np.random.default_rng(seed)
    bgp = plan7.Background(abc).residue_frequencies.astype(np.float64)
    bgp /= bgp.sum()
    mat = np.asarray(hmm.match_emissions, dtype=np.float64)
    ins = np.asarray(hmm.insert_emissions, dtype=np.float64)
    t = np.asarray(hmm.transition_probabilities, dtype=np.float64)
    cmat, cins = np.cumsum(mat, axis=1), np.cumsum(ins, axis=1)
    ct = np.zeros((hmm.M + 1, 4))
    s3 = np.maximum(t[:, 0] + t[:, 1] + t[:, 2], 1e-30)
    ct[:, 0], ct[:, 1] = t[:, 0] / s3, (t[:, 0] + t[:, 1]) / s3
    ct[:, 2] = t[:, 3] / np.maximum(t[:, 3] + t[:, 4], 1e-30)
    ct[:, 3] = t[:, 5] / np.maximum(t[:, 5] + t[:, 6], 1e-30)
    flank = lambda lo, hi: rng.choice(abc.K, size=int(rng.integers(lo, hi)), p=bgp).astype(np.uint8)
    seqs = [easel.DigitalSequence(abc, name=f"bg{i}", sequence=flank(50, 600)) for i in range(nbg)]
    for h in range(nhom):
        parts = []
        for d in range(1 if rng.random() < 0.7 else int(rng.integers(2, 4))):
            dom = np.array(bench.emit_from_model(hmm, rng, (cmat, cins, ct)), dtype=np.uint8)
            lo = int(rng.integers(0, max(1, len(dom) // 3)))
            hi = int(rng.integers(max(lo + 10, 2 * len(dom) // 3), len(dom) + 1))
            parts += [flank(5, 120), dom[lo:hi]]
        parts.append(flank(5, 120))
        seqs.append(easel.DigitalSequence(abc, name=f"hom{h}", sequence=np.concatenate(parts)))
    return easel.DigitalSequenceBlock(abc, seqs)


@pytest.mark.parametrize("model", ["PF02826", "KR", "Thioesterase", "LuxC", 30, 150, 400, "dna 120", "RF00001"])
def test_host_stage_agrees_with_the_oracle_on_single_domain_regions(oracle, model):
    """The product's host stage and the oracle get the same parser rows (the oracle's) for 100 background + 200 homolog
    targets; for every region the oracle resolves: the product defines a domain with the same envelope, the same alignment
    and model coordinates, and score and bias to the last bits (TOL_BITS): both sides sum in upstream's order."""
    if model == "dna 120":
        hmm = random_hmm(120, seed=620, alphabet=easel.Alphabet.dna(), conserved=0.3)
    else:
        hmm = load_hmms(model)[0] if isinstance(model, str) else random_hmm(model, seed=500 + model)
    block = _homolog_block(hmm, 100, 200, seed=11)
    pli = plan7.Pipeline(hmm.alphabet, E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
    hits = host_pipeline.host_search(oracle, hmm, block, pipeline=pli)
    op = oracle.OracleProfile(hmm, pli.background, 400)
    by_name = {s.name: s for s in block}
    envelopes = differing = 0
    for h in hits:
        envs, counts = oracle.domains_single(op, np.asarray(by_name[h.name].sequence, dtype=np.uint8))
        assert h.nregions == counts[0]
        prod = {(d.env_from, d.env_to): d for d in h.domains}
        for e in envs:
            envelopes += 1
            d = prod.get((int(e[0]), int(e[1])))
            assert d is not None, (h.name, e[:2], sorted(prod))
            a = d.alignment
            if (a.target_from, a.target_to, a.hmm_from, a.hmm_to) != tuple(int(v) for v in e[2:6]):
                differing += 1
            else:
                assert abs(d.score - e[9]) <= TOL_BITS and abs(d.bias - e[10]) <= TOL_BITS, (h.name, d.score, e[9], d.bias, e[10])
                assert abs(d.envelope_score * np.log(2.0) - e[6]) <= 1e-6 * max(1.0, abs(e[6]))
                assert abs(d.accuracy - e[8] / (1.0 + e[1] - e[0])) <= 1e-6                    # the optimal-accuracy score per envelope residue
    assert envelopes >= 150
    assert differing == 0, (differing, envelopes)                           # one summation order on both sides: upstream's


@pytest.mark.parametrize("model", ["PF02826", "KR", "LuxC"])
def test_host_stage_agrees_with_the_oracle_on_ensemble_regions(oracle, model):
    """Targets with two or three homologous fragments: regions that hold several domains go through the ensemble of 200
    sampled tracebacks on both sides.  The samples are the same -- same generator, same draws, same Forward matrix bit for
    bit (upstream's striped summation order on both sides), hence the same choices.  Required: every target has identical
    domain lists (all coordinates) with scores and biases to the last bits and the same (nregions, nclustered, noverlaps,
    nenvelopes)."""
    hmm = load_hmms(model)[0]
    block = _homolog_block(hmm, 50, 200, seed=21)
    pli = plan7.Pipeline(hmm.alphabet, E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
    hits = host_pipeline.host_search(oracle, hmm, block, pipeline=pli)
    op = oracle.OracleProfile(hmm, pli.background, 400)
    by_name = {s.name: s for s in block}
    with_ensembles = same = 0
    for h in hits:
        envs, counts = oracle.domains(op, np.asarray(by_name[h.name].sequence, dtype=np.uint8))
        ours = [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]
        theirs = [tuple(int(v) for v in e[:6]) for e in envs]
        assert h.nregions == counts[0]
        if counts[2] == 0:
            assert ours == theirs, (h.name, ours, theirs)
            continue
        with_ensembles += 1
        if ours == theirs:
            same += 1
            assert (h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) == (counts[0], counts[2], counts[4], counts[1]), h.name
            for e, d in zip(envs, h.domains):
                assert abs(d.score - e[9]) <= TOL_BITS and abs(d.bias - e[10]) <= TOL_BITS, (h.name, d.score, e[9], d.bias, e[10])
    assert with_ensembles >= 10 and same == with_ensembles, (same, with_ensembles)


def test_oracle_sequence_scores_reproduce_the_reference_target_tables(oracle, proteome):
    """PF02826.tbl and RREFam.tbl: the full-sequence score, bias and E-value of every reported target from the oracle's own
    domain definition (null2 correction over all residues, or the sum of the domains when that is higher: p7_pipeline.c)."""
    by_name = {s.name: s for s in proteome}
    checked = 0
    for hmm in load_hmms("PF02826") + load_hmms("RREFam"):
        rows = golden_table("PF02826.tbl" if hmm.name == "2-Hacid_dh_C" else "RREFam.tbl", hmm.name)
        if not rows:
            continue
        op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
        for r in rows:
            envs, counts, sq = oracle.domains(op, np.asarray(by_name[r[0]].sequence, dtype=np.uint8), want_sequence=True)
            assert abs(sq["score"] - float(r[5])) <= 0.051, (hmm.name, r[0], r[5], sq)
            assert abs((sq["pre_score"] - sq["score"]) - float(r[6])) <= 0.051, (hmm.name, r[0], r[6], sq)
            evalue = float(np.exp(sq["lnP"])) * len(proteome)
            assert abs(evalue - float(r[4])) <= 0.06 * float(r[4]), (hmm.name, r[0], r[4], evalue)
            assert int(sq["ndom"]) >= int(r[17])                      # the table's "dom" column counts the domains it reports
            checked += 1
    assert checked >= 30


def test_degenerate_residues_inside_envelopes(oracle):
    """Targets whose homologous fragments carry X, B and Z residues (the null2 odds of a degenerate code are the plain mean
    over its residues, esl_abc_FAvgScVec; its emission odds come from the profile's own rows): host stage == oracle."""
    hmm = load_hmms("PF02826")[0]
    block = _homolog_block(hmm, 20, 120, seed=31)
    rng = np.random.default_rng(5)
    seqs = []
    for s in block:
        a = np.asarray(s.sequence, dtype=np.uint8).copy()
        hit = rng.random(len(a)) < 0.03
        a[hit] = rng.choice([26, 21, 23], size=int(hit.sum()))            # X, B, Z
        seqs.append(easel.DigitalSequence(hmm.alphabet, name=s.name, sequence=a))
    block = easel.DigitalSequenceBlock(hmm.alphabet, seqs)
    pli = plan7.Pipeline(hmm.alphabet, E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
    hits = host_pipeline.host_search(oracle, hmm, block, pipeline=pli)
    op = oracle.OracleProfile(hmm, pli.background, 400)
    by_name = {s.name: s for s in block}
    compared = 0
    for h in hits:
        envs, counts = oracle.domains(op, np.asarray(by_name[h.name].sequence, dtype=np.uint8))
        if counts[2]:
            continue                                                     # ensemble regions: see the test above
        ours = [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]
        assert ours == [tuple(int(v) for v in e[:6]) for e in envs], h.name
        for e, d in zip(envs, h.domains):
            assert abs(d.score - e[9]) <= TOL_BITS and abs(d.bias - e[10]) <= TOL_BITS, (h.name, d.score, e[9], d.bias, e[10])
            compared += 1
    assert compared >= 80


def test_alignment_display_lines(oracle, proteome):
    """p7_alidisplay_Create restated in the oracle: the model, match, sequence and posterior lines of the reference's own
    Thioesterase answer (reference tests/test_hmmer.py:51-106), and the product's host stage against it on every domain of
    170 synthetic targets of two models (trailing delete states of an optimal-accuracy trace are not displayed; inserts are
    lower case; '+' marks emission odds above 1)."""
    hmm = load_hmms("Thioesterase")[0]
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    seq = np.asarray([s for s in proteome if s.name == "938293.PRJEB85.HG003687_113"][0].sequence, dtype=np.uint8)
    assert oracle.domain_alignment(op, seq, 115, 129) == ("GWSfGGvlAyEmArq", "G+S+GG +A ++A++", "GHSMGGSVAVAIAHE", "9************96")
    compared = with_inserts = with_deletes = 0
    for name in ("PF02826", "KR"):
        hmm = load_hmms(name)[0]
        block = _homolog_block(hmm, 20, 150, seed=41)
        pli = plan7.Pipeline(hmm.alphabet, E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
        hits = host_pipeline.host_search(oracle, hmm, block, pipeline=pli)
        op = oracle.OracleProfile(hmm, pli.background, 400)
        by_name = {s.name: s for s in block}
        for h in hits:
            seq = np.asarray(by_name[h.name].sequence, dtype=np.uint8)
            for d in h.domains:
                a = d.alignment
                lines = oracle.domain_alignment(op, seq, d.env_from, d.env_to)
                assert lines == (a.hmm_sequence, a.identity_sequence, a.target_sequence, a.posterior_probabilities), (name, h.name, d.env_from)
                compared += 1
                with_inserts += "." in lines[0]
                with_deletes += "-" in lines[2]
    assert compared >= 300 and with_inserts >= 20 and with_deletes >= 20


def _oracle_search(oracle, hmm, block, E=10.0, domE=10.0, incE=0.01, incdomE=0.01, T=None, domT=None, incT=None, incdomT=None, cutoffs=None):
    """A whole hmmsearch through the oracle alone: filter cascade, domain definition, sequence scores, then the reference's
    reporting logic restated here -- p7_pli_TargetReportable per target as it is found, sort by E-value (ties by name),
    p7_tophits_Threshold: targets reported / included by E-value over Z = number of targets, domains by E-value over domZ =
    number of reported targets, a domain included only if its target is.  T / domT / incT / incdomT: score thresholds
    instead of E-values; cutoffs = (sequence, domain): the model's bit cutoffs for reporting and inclusion alike
    (p7_pli_NewModelThresholds)."""
    if cutoffs is not None:
        T = incT = cutoffs[0]
        domT = incdomT = cutoffs[1]
    bg = plan7.Background(hmm.alphabet)
    op = oracle.OracleProfile(hmm, bg, 400)
    recs, ctr = op.cascade_block(block.packed())
    Z = len(block)
    found = []
    for t in range(Z):
        if recs[t].stage != 4:
            continue
        envs, counts, sq = oracle.domains(op, np.asarray(block[t].sequence, dtype=np.uint8), want_sequence=True)
        if counts[0] == 0 or len(envs) == 0:
            continue
        if (sq["score"] >= T) if T is not None else (np.exp(sq["lnP"]) * Z <= E):
            found.append((block[t].name, sq, envs))
    by_score = incT is not None                   # the sort key is the score when inclusion goes by score (p7_pipeline.c)
    found.sort(key=(lambda h: (-h[1]["score"], h[0])) if by_score else (lambda h: (h[1]["lnP"], h[0])))
    reported = [(sq["score"] >= T) if T is not None else (np.exp(sq["lnP"]) * Z <= E) for _, sq, _ in found]
    domZ = sum(reported)
    out = []
    for (name, sq, envs), rep in zip(found, reported):
        inc = rep and ((sq["score"] >= incT) if incT is not None else (np.exp(sq["lnP"]) * Z <= incE))
        doms = [(bool(rep and ((e[9] >= domT) if domT is not None else (np.exp(e[11]) * domZ <= domE))),
                 bool(inc and ((e[9] >= incdomT) if incdomT is not None else (np.exp(e[11]) * domZ <= incdomE))),
                 float(np.exp(e[11]) * domZ), float(np.exp(e[11]) * Z)) for e in envs]
        out.append((name, bool(rep), bool(inc), float(sq["score"]), float(np.exp(sq["lnP"]) * Z), doms))
    return out


@pytest.mark.parametrize("model", ["PF02826", "RREFam", "KR", "LuxC", "Thioesterase"])
def test_whole_search_through_the_oracle_alone_equals_the_host_pipeline(oracle, proteome, model):
    """Hit lists of the fixture models against the proteome: the oracle's own search (no product code) and the product's
    host pipeline (oracle filters + product host stage and hit list) name the same targets in the same order with the same
    reported / included flags for targets and domains, scores to 2e-3 bit and E-values to 0.2 %."""
    compared = 0
    for hmm in load_hmms(model):
        want = _oracle_search(oracle, hmm, proteome)
        hits = host_pipeline.host_search(oracle, hmm, proteome)
        got = [(h.name, h.reported, h.included, h.score, h.evalue, [(d.reported, d.included, d.c_evalue, d.i_evalue) for d in h.domains]) for h in hits]
        assert [g[:3] for g in got] == [w[:3] for w in want], hmm.name
        for g, w in zip(got, want):
            assert abs(g[3] - w[3]) <= TOL_BITS and abs(g[4] - w[4]) <= 1e-4 * w[4] + 1e-300, (hmm.name, g[0], g[3:5], w[3:5])
            assert [d[:2] for d in g[5]] == [d[:2] for d in w[5]], (hmm.name, g[0])
            for dg, dw in zip(g[5], w[5]):
                assert abs(dg[2] - dw[2]) <= 1e-4 * dw[2] + 1e-300 and abs(dg[3] - dw[3]) <= 1e-4 * dw[3] + 1e-300, (hmm.name, g[0], dg, dw)
            compared += 1
    assert compared >= 1


@pytest.mark.parametrize("M", [1, 2, 3, 4, 7, 13])
def test_models_of_a_few_nodes_against_targets_of_a_few_residues(oracle, M):
    """300 targets of 1 - 59 residues, a third of them with tandem copies of the model's consensus, some all X: the same hits
    and the same domain lists from the host pipeline and from the oracle alone.  (With three nodes the regions of tandem
    copies are resolved by sampled tracebacks whose domains are too short ever to link -- upstream counts the overlap on
    the model without the + 1 -- so they yield no envelope at all: both sides agree on that, too.)"""
    abc = easel.Alphabet.amino()
    rng = np.random.default_rng(9 + M)
    hmm = random_hmm(M, seed=100 + M, conserved=0.9)
    cons = np.argmax(hmm.match_emissions[1:], axis=1).astype(np.uint8)
    seqs = []
    for t in range(300):
        L = int(rng.integers(1, 60))
        a = rng.integers(0, 20, size=L).astype(np.uint8)
        if t % 3 == 0:
            rep = np.tile(cons, int(rng.integers(1, 6)))[:L]
            p0 = int(rng.integers(0, max(1, L - len(rep) + 1)))
            a[p0:p0 + len(rep)] = rep[:L - p0]
        if t % 17 == 0:
            a[:] = 26
        seqs.append(easel.DigitalSequence(abc, name=f"t{t}", sequence=a))
    block = easel.DigitalSequenceBlock(abc, seqs)
    loose = dict(E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
    want = _oracle_search(oracle, hmm, block, **loose)
    hits = host_pipeline.host_search(oracle, hmm, block, pipeline=plan7.Pipeline(abc, **loose))
    assert [h.name for h in hits] == [w[0] for w in want]
    op = oracle.OracleProfile(hmm, plan7.Background(abc), 400)
    by_name = {s.name: s for s in block}
    for h in hits:
        envs, counts = oracle.domains(op, np.asarray(by_name[h.name].sequence, dtype=np.uint8))
        ours = [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]
        assert ours == [tuple(int(v) for v in e[:6]) for e in envs], (M, h.name)
        assert (h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) == (counts[0], counts[2], counts[4], counts[1]), (M, h.name)
    if M >= 3:
        assert len(hits) >= 20


def test_score_thresholds_and_bit_cutoffs_through_the_oracle_alone(oracle, proteome):
    """The same whole-search comparison with score thresholds (T, domT, incT, incdomT) and with the model's gathering and
    trusted cutoffs: targets, order and every reported / included flag."""
    def flags(hits):
        return [(h.name, h.reported, h.included, [(d.reported, d.included) for d in h.domains]) for h in hits]
    hmm = load_hmms("PF02826")[0]
    for kw in (dict(T=20.0, domT=10.0, incT=50.0, incdomT=30.0), dict(T=-5.0), dict(incT=100.0, incdomT=100.0)):
        want = _oracle_search(oracle, hmm, proteome, **kw)
        got = host_pipeline.host_search(oracle, hmm, proteome, pipeline=plan7.Pipeline(hmm.alphabet, **kw))
        assert flags(got) == [(w[0], w[1], w[2], [d[:2] for d in w[5]]) for w in want], kw
    for which, pair in (("gathering", hmm.cutoffs.gathering), ("trusted", hmm.cutoffs.trusted)):
        want = _oracle_search(oracle, hmm, proteome, cutoffs=pair)
        got = host_pipeline.host_search(oracle, hmm, proteome, pipeline=plan7.Pipeline(hmm.alphabet, bit_cutoffs=which))
        assert flags(got) == [(w[0], w[1], w[2], [d[:2] for d in w[5]]) for w in want], which
        assert len(got) >= 5
