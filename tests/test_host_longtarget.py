"""CPU tests of the long-target (nhmmer) path: the oracle's sequential SSV scan seeds the windows, the product's host
tail (p7x_longtarget_from_seeds: window filters, long-target Viterbi, parsers, domain definition with long_target = TRUE,
hit construction, p7_tophits_ComputeNhmmerEvalues / RemoveDuplicates) does the rest.  Pinned by the reference's own
nhmmer fixtures (reference tests/test_hmmer.py:631-795): tables/bmyD1.tbl and bmyD2.tbl (real nhmmer 3.3 / 3.4 output)
and the RF00001 known answers."""
import numpy as np
import pytest

import host_pipeline
from conftest import GOLDEN, golden_table, load_hmms
from pyhmmer_amd import easel, plan7


def _read(name, alphabet):
    with easel.SequenceFile(GOLDEN / "seqs" / name, digital=True, alphabet=alphabet) as f:
        return f.read_block()


def _rows(hits):
    return [(h.name, d.alignment.hmm_from, d.alignment.hmm_to, d.alignment.target_from, d.alignment.target_to, d.env_from, d.env_to,
             d.strand, h.evalue, h.score, d.bias, h.reported, h.included) for h in hits for d in [h.domains[0]]]


def rounds_to(value, text):
    """<value> printed with as many decimals as the table column <text> gives the table's number (half a unit of the last
    printed digit plus float noise)."""
    decimals = len(text.split(".")[1]) if "." in text else 0
    return abs(value - float(text)) <= 0.5 * 10.0 ** -decimals + 1e-4


def check_nhmmer_table(hits, rows):
    """reference TestNhmmer.assertTableEqual (test_hmmer.py:673-690 compares best-domain bias, score and i_evalue to
    0.1) tightened to the table's own print precision -- score and bias to the printed decimal, E-value to its two
    significant digits -- plus every coordinate and strand column, exactly, for every row."""
    reported = [h for h in hits if h.reported]
    assert len(reported) == len(rows)
    for row, hit in zip(rows, reported):
        d = hit.best_domain
        assert hit.name == row[0]
        assert (hit.accession or "-") == row[1]
        assert (d.alignment.hmm_from, d.alignment.hmm_to) == (int(row[4]), int(row[5]))
        assert (d.alignment.target_from, d.alignment.target_to) == (int(row[6]), int(row[7]))
        assert (d.env_from, d.env_to) == (int(row[8]), int(row[9]))
        assert hit.length == int(row[10]) and d.strand == row[11]
        assert rounds_to(d.bias, row[14]), (d.bias, row[14])
        assert rounds_to(d.score, row[13]), (d.score, row[13])
        assert d.i_evalue == pytest.approx(float(row[12]), abs=0.1)
        if float(row[12]) > 0:
            assert float("%.2g" % d.i_evalue) == pytest.approx(float(row[12]), rel=0.045), (d.i_evalue, row[12])   # one unit of the second digit


def test_genbank_targets_are_read(libp7x):
    seqs = _read("BGC0001090.gbk", easel.Alphabet.dna())
    assert len(seqs) == 1 and seqs[0].name == "BGC0001090" and seqs[0].accession == "BGC0001090.1" and len(seqs[0]) == 44660
    assert seqs[0].description.startswith("Bacillus amyloliquefaciens subsp. plantarum str. FZB42, complete")


def test_bmyd_hmm_vs_bgc_matches_nhmmer_table(libp7x, oracle):
    """reference test_bmyd_hmm_bgc_block / _file (test_hmmer.py:734-753) against tables/bmyD1.tbl: both rows exactly
    (coordinates, strand, score, bias, E-value)."""
    hmm = load_hmms("bmyD")[0]
    seqs = _read("BGC0001090.gbk", hmm.alphabet)
    hits = host_pipeline.host_nhmmer(oracle, hmm, seqs)
    assert hits.long_targets and hits.strand is None and hits.block_length == 0x40000
    assert len(hits.reported) == 2
    check_nhmmer_table(hits, golden_table("bmyD1.tbl"))
    assert [h.domains[0].strand for h in hits.reported] == ["+", "-"]
    # one strand at a time
    for strand, want in (("watson", "+"), ("crick", "-")):
        one = host_pipeline.host_nhmmer(oracle, hmm, seqs, plan7.LongTargetsPipeline(hmm.alphabet, strand=strand))
        assert one.strand == strand and [h.domains[0].strand for h in one.reported] == [want]
        # E-values refer to the residues searched: half of them with one strand
        both = next(h for h in hits.reported if h.domains[0].strand == want)
        assert one.reported[0].score == pytest.approx(both.score, abs=1e-3)


def test_bmyd_hmm_vs_genome_matches_nhmmer_table(libp7x, oracle):
    """reference test_bmyd_hmm_genome_block / _file (test_hmmer.py:755-775) against tables/bmyD2.tbl (391 kb contig, two
    blocks of 0x40000 with max_length overlap): three reported hits, two of them included, in the table's order, every
    coordinate of every row exact, scores / biases / E-values to the table's print precision (the long-target envelope
    scoring is derived, DESIGN.md section 3.8: length model of the envelope's own length, composition-adjusted background,
    bias = the score lost to the adjustment)."""
    hmm = load_hmms("bmyD")[0]
    seqs = _read("1390.SAMEA104415756.OFHT01000022.fna", hmm.alphabet)
    hits = host_pipeline.host_nhmmer(oracle, hmm, seqs)
    check_bmyd2_table(hits, golden_table("bmyD2.tbl"))


def check_bmyd2_table(hits, rows):
    assert len(hits.reported) == 3 and len(hits.included) == 2
    check_nhmmer_table(hits, rows)


def test_rf00001_known_answers(libp7x, oracle):
    """reference test_rf0001_genome_file / _wlen_3878 (test_hmmer.py:777-795)."""
    hmm = load_hmms("RF00001")[0]
    seqs = _read("1390.SAMEA104415756.OFHT01000024.fna", hmm.alphabet)
    hits = host_pipeline.host_nhmmer(oracle, hmm, seqs)
    assert len(hits) == 1
    assert float("%.2g" % hits[0].evalue) == pytest.approx(2.5e-17, rel=1e-6) and hits[0].best_domain.strand == "-"
    hits = host_pipeline.host_nhmmer(oracle, hmm, seqs, plan7.LongTargetsPipeline(hmm.alphabet, window_length=3878))
    assert len(hits) == 2
    assert float("%.2g" % hits[0].evalue) == pytest.approx(5.4e-17, rel=1e-6) and hits[1].evalue == pytest.approx(0.3, abs=0.005)
    assert hits[0].best_domain.strand == "-" and hits[1].best_domain.strand == "-"


def test_long_targets_pipeline_arguments(libp7x):
    """reference plan7.pyx:7062-7130: nucleotide alphabets only, strand names, window parameters."""
    dna = easel.Alphabet.dna()
    with pytest.raises(ValueError):
        plan7.LongTargetsPipeline(easel.Alphabet.amino())
    with pytest.raises(Exception):
        plan7.LongTargetsPipeline(dna, strand="both")
    with pytest.raises(Exception):
        plan7.LongTargetsPipeline(dna, window_length=2)
    pli = plan7.LongTargetsPipeline(dna, strand="crick", B1=110, block_length=4096)
    c = pli._cfg()
    assert (c.long_targets, c.strands, c.B1, c.B2, c.B3, c.block_length, c.F2, c.F3) == (1, 2, 110, 240, 1000, 4096, 3e-3, 3e-5)


def test_builder_max_length_reproduces_the_fixtures_maxl(libp7x):
    """p7_Builder_MaxLength (reference plan7.pyx:7346-7354 calls it with window_beta): with hmmbuild's default
    beta = 1e-7 the restatement gives the MAXL line hmmbuild wrote into both nucleotide fixtures; a looser beta gives a
    shorter window, a tighter one a longer."""
    import ctypes as C
    from pyhmmer_amd import _lib
    got = {}
    for name, want in (("RF00001", 305), ("bmyD", 1736)):
        hmm = load_hmms(name)[0]
        assert hmm.max_length == want
        view, keep = hmm._view()
        for beta in (1e-7, 1e-3, 1e-9):
            w = C.c_int32()
            assert _lib.lib().p7x_hmm_max_length(C.byref(view), beta, C.byref(w)) == 0
            got[(name, beta)] = w.value
        assert got[(name, 1e-7)] == want
        assert got[(name, 1e-3)] < want < got[(name, 1e-9)]
    w = C.c_int32()
    assert _lib.lib().p7x_hmm_max_length(C.byref(view), 0.0, C.byref(w)) != 0


def test_window_beta_sets_the_evalue_window(libp7x):
    """LongTargetsPipeline(window_beta=...) with an HMM query (plan7.pyx:7346-7354): the E-values are computed for
    p7_Builder_MaxLength(hmm, window_beta), the scan keeps the profile's own max_length, and -- as in the reference -- the
    HMM's max_length is replaced; window_length overrides both."""
    hmm = load_hmms("bmyD")[0]
    cfg = plan7.LongTargetsPipeline(hmm.alphabet, window_beta=1e-3)._cfg()
    pli = plan7.LongTargetsPipeline(hmm.alphabet, window_beta=1e-3)
    om = pli._windowed_om(hmm, 400, cfg)
    assert om._info.max_length == 1736 and 1203 < cfg.evalue_window_length < 1736 and hmm.max_length == cfg.evalue_window_length
    hmm = load_hmms("bmyD")[0]
    hmm.max_length = None                                     # no MAXL line: the computed bound serves the scan as well
    cfg = pli._cfg()
    om = pli._windowed_om(hmm, 400, cfg)
    assert om._info.max_length == cfg.evalue_window_length == hmm.max_length
    cfg = plan7.LongTargetsPipeline(hmm.alphabet, window_length=3878)._cfg()
    assert cfg.window_length == 3878 and cfg.evalue_window_length == -1


def test_search_dealt_over_parts_equals_the_whole(libp7x, oracle):
    """nhmmer over several devices (cfg.lt_part / lt_nparts + p7x_tophits_merge_longtargets): the (target, block, strand)
    units are independent, so a search dealt over 2, 3 or 4 parts -- E-values for all residues, duplicates across block
    boundaries and thresholds done once for the merged list -- equals the one-part search, field by field."""
    hmm = load_hmms("bmyD")[0]
    seqs = _read("1390.SAMEA104415756.OFHT01000022.fna", hmm.alphabet)          # 391 kb: two blocks x two strands
    # a short block length gives the parts more units and puts hits next to block boundaries
    for block_length in (0x40000, 60000):
        mk = lambda: plan7.LongTargetsPipeline(hmm.alphabet, block_length=block_length)
        whole = host_pipeline.host_nhmmer(oracle, hmm, seqs, mk())
        assert whole._nunits == 2 * len(host_pipeline.blocks_of(len(seqs[0]), block_length, hmm.max_length))
        for nparts in (2, 3, 4, whole._nunits):
            parts = host_pipeline.host_nhmmer(oracle, hmm, seqs, mk(), nparts=nparts)
            assert _rows(parts) == _rows(whole), (block_length, nparts)
            assert parts.stage_counts == whole.stage_counts and parts.searched_residues == whole.searched_residues
            assert [(h.evalue, h.score, h.reported, h.included, h.duplicate) for h in parts] == \
                   [(h.evalue, h.score, h.reported, h.included, h.duplicate) for h in whole]
    check_bmyd2_table(host_pipeline.host_nhmmer(oracle, hmm, seqs, nparts=3), golden_table("bmyD2.tbl"))
    # a part cannot pass for a finished hit list, and parts of different searches do not merge
    import ctypes as C
    from pyhmmer_amd import _lib
    bad = (C.c_void_p * 1)(None)
    out = C.c_void_p()
    assert _lib.lib().p7x_tophits_merge_longtargets(bad, 1, C.byref(out)) != 0


def test_long_target_tail_against_the_oracle_restatement(oracle):
    """The product's host tail (fed with the oracle's SSV seeds) against oracle/p7_oracle_lt.c on a 300 kbp synthetic chromosome with
    copies across block seams, short fragments and low-complexity stretches: the numbers of windows past every filter are equal,
    every hit lies inside a window the oracle lets through Forward, every score follows from the hit's own coordinates by
    the oracle's scoring rule (1e-3 bit), and the coordinates are those of the oracle's own domain definition.  The 2 Mbp version of this test runs the device path (tests/test_gpu_longtarget.py)."""
    import lt_oracle_check as lc
    hmm = load_hmms("bmyD")[0]
    pli = plan7.LongTargetsPipeline(hmm.alphabet, block_length=65536)
    seq = lc.synthetic_chromosome(hmm, 300_000, seed=11, block_length=pli.block_length, max_length=hmm.max_length)
    block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chrT", sequence=seq)])
    hits = host_pipeline.host_nhmmer(oracle, hmm, block, pipeline=pli)
    windows = lc.oracle_windows(pli, hmm, seq)
    nwin, nshort = lc.check_hits_against_oracle(pli, hmm, seq, hits, min_windows=100, min_short=3, oracle=windows)
    assert len(hits) >= 20
    # ... and the coordinates themselves from the oracle's own long-target domain definition (p7o_lt_domains: regions,
    # ensembles, the envelope under its own length model with composition-adjusted emissions, cut back to its alignment):
    # every hit's envelope, alignment and model coordinates, exactly
    checked, clustered = lc.check_hit_coordinates_against_oracle(pli, hmm, seq, hits, oracle=windows)
    assert checked == len(hits) and clustered >= 20


def _cut_model(full, lo, hi, name):
    """Nodes lo+1 .. hi of <full> as a model of their own (test input: a short nucleotide model, short windows)."""
    abc = full.alphabet
    h = plan7.HMM(abc, hi - lo, name)
    h.transition_probabilities[1:] = full.transition_probabilities[lo + 1:hi + 1]
    h.transition_probabilities[0] = full.transition_probabilities[0]
    h.match_emissions[1:] = full.match_emissions[lo + 1:hi + 1]
    h.insert_emissions[:] = full.insert_emissions[lo:hi + 1]
    t = np.array(h.transition_probabilities[hi - lo])
    t[0], t[2] = t[0] + t[2], 0.0
    t[5], t[6] = 1.0, 0.0
    h.transition_probabilities[hi - lo] = t
    h.composition = full.composition
    h.consensus = full.consensus[lo:hi]
    h._evparam[:] = full._evparam
    return h


def test_short_windows_against_the_oracle_restatement(oracle):
    """A 40-node model: its windows are a few dozen residues long, so the background an envelope is rescored against is mixed
    with smoothing 25 / max(50, n) > 0.25 -- the branch of upstream's reparameterize_model that no fixture of the reference
    reaches (all its windows are longer than 100 residues)."""
    import lt_oracle_check as lc
    full = load_hmms("bmyD")[0]
    hmm = _cut_model(full, 400, 440, "bmyD_40")
    pli = plan7.LongTargetsPipeline(hmm.alphabet, block_length=32768, E=100.0, window_length=70)     # nhmmer --w_length 70
    seq = lc.synthetic_chromosome(hmm, 120_000, seed=5, block_length=pli.block_length, max_length=70, long_copies=False)
    block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chrS", sequence=seq)])
    hits = host_pipeline.host_nhmmer(oracle, hmm, block, pipeline=pli)
    windows = lc.oracle_windows(pli, hmm, seq)
    nwin, nshort, nshortwin = lc.check_hits_against_oracle(pli, hmm, seq, hits, min_windows=20, min_short=5, want_short_windows=True, oracle=windows)
    assert nshortwin >= 3 and len(hits) >= 5
    assert lc.check_hit_coordinates_against_oracle(pli, hmm, seq, hits, oracle=windows)[0] >= len(hits) - 2


@pytest.mark.parametrize("name,M", [("bmyD", None), ("RF00001", None), (None, 60), (None, 638), (None, 1278), (None, 2047)])
def test_ssv_scan_tables_against_the_oracle_profile(libp7x, oracle, name, M):
    """The table the long-target SSV kernel stages in LDS (test seam p7x_debug_ssv_tables; host code): every packed pair is
    bias - rb[x][k] of the oracle's pressed-format byte costs at the documented place ([parity][x][quad][lane][c], register
    j = 4 quad + c of the lane = nodes (2g - 1, 2g) / (2g, 2g + 1), g = lane R + j), padding is -512, the virtual node M + 1
    of the every-second-row flavour has emission 0, R is the smallest register count that holds the model, and the reported
    one-row loss is the largest canonical cost above the bias."""
    import ctypes as C
    from conftest import random_hmm
    from pyhmmer_amd import _lib
    hmm = load_hmms(name)[0] if name else random_hmm(M, seed=4000 + M, alphabet=easel.Alphabet.dna())
    bg = plan7.Background(hmm.alphabet)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    op = oracle.OracleProfile(hmm, bg, 400)
    M, Q, bias = hmm.M, int(op.p.Q16), int(op.p.bias_b)
    rbv = np.asarray(op.arr("rbv")).astype(np.int64)
    cost = np.zeros((4, M + 3), dtype=np.int64)                     # un-striped: node k sits at vector (k-1) % Q, byte (k-1) // Q
    for k in range(1, M + 1):
        cost[:, k] = rbv[:4, ((k - 1) % Q) * 16 + (k - 1) // Q]
    sval = bias - cost                                               # the signed emission the kernel adds
    for pair in (0, 1):
        R, slack = C.c_int32(), C.c_int32()
        n = _lib.lib().p7x_debug_ssv_tables(om._handle, pair, C.byref(R), C.byref(slack), None, 0)
        assert n > 0, _lib.last_error()
        tab = np.zeros(n, dtype=np.uint32)
        assert _lib.lib().p7x_debug_ssv_tables(om._handle, pair, C.byref(R), C.byref(slack), tab.ctypes.data, n) == n
        R = R.value
        top = M + pair
        need = (top + 1) // 2 + 1
        assert R * 64 >= need and (R == 2 or [r for r in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24, 32, 48) if r * 64 >= need][0] == R)
        assert slack.value == int((cost[:, 1:M + 1] - bias).max())
        R4 = (R + 3) // 4
        assert n == 2 * 4 * R4 * 64 * 4
        t = tab.reshape(2, 4, R4, 64, 4)
        lo = (t & 0xffff).astype(np.int64); lo[lo >= 32768] -= 65536
        hi = (t >> 16).astype(np.int64); hi[hi >= 32768] -= 65536

        def want(x, k):
            if pair and k == M + 1:
                return 0
            return int(sval[x, k]) if 1 <= k <= M else -512
        rng = np.random.default_rng(M)
        lanes = np.unique(np.concatenate([[0, 1, 63], rng.integers(0, 64, size=12)]))
        for par in (0, 1):
            for x in range(4):
                for lane in lanes:
                    for j in range(R):
                        g = int(lane) * R + j
                        k0 = 2 * g - 1 if par == 0 else 2 * g
                        assert (int(lo[par, x, j // 4, lane, j % 4]), int(hi[par, x, j // 4, lane, j % 4])) == (want(x, k0), want(x, k0 + 1)), (pair, par, x, lane, j)
                    for j in range(R, 4 * R4):                       # the unused registers of the last quad
                        assert t[par, x, j // 4, lane, j % 4] == 0


@pytest.mark.parametrize("model,target,table", [("bmyD", "BGC0001090.gbk", "bmyD1.tbl"), ("bmyD", "1390.SAMEA104415756.OFHT01000022.fna", "bmyD2.tbl"),
                                                ("RF00001", "1390.SAMEA104415756.OFHT01000024.fna", None)])
def test_oracle_long_target_domains_reproduce_the_nhmmer_fixtures(oracle, model, target, table):
    """The reference's nhmmer answers from the oracle alone: its windows (p7_oracle_lt.c), its own long-target domain
    definition inside them (p7_oracle_dd.c: p7o_lt_domains) and its scoring rule give every row of bmyD1.tbl / bmyD2.tbl --
    hmm, alignment and envelope coordinates exactly, score and bias at print precision -- and the RF00001 hit."""
    import lt_oracle_check as lc
    hmm = load_hmms(model)[0]
    seqs = _read(target, hmm.alphabet)
    pli = plan7.LongTargetsPipeline(hmm.alphabet)
    rows = golden_table(table) if table else None
    found = 0
    for s in seqs:
        seq = np.asarray(s.sequence, dtype=np.uint8)
        op, max_length, units, total = lc.oracle_windows(pli, hmm, seq)
        doms = []
        for (i, n, strand), win in units.items():
            blk = seq[i:i + n] if strand == 0 else host_pipeline.DNA_COMP[seq[i:i + n][::-1]]
            for ws, wl in [(int(w[0]), int(w[1])) for w in win]:
                envs, counts = oracle.lt_domains(op, blk[ws - 1:ws - 1 + wl], seed=pli.seed)
                for e in envs:
                    pos = (lambda x: i + ws - 1 + int(x)) if strand == 0 else (lambda x: i + n - (ws - 1 + int(x)) + 1)
                    score, bias, lnp = oracle.lt_domain_score(op, max_length, int(e[1] - e[0] + 1), int(e[3] - e[2] + 1), e[6], e[7], pli.null2)
                    doms.append((s.name, int(e[4]), int(e[5]), pos(e[2]), pos(e[3]), pos(e[0]), pos(e[1]), "+" if strand == 0 else "-", score, bias))
        if rows:
            for r in rows:
                if r[0] != s.name:
                    continue
                want = (r[0], int(r[4]), int(r[5]), int(r[6]), int(r[7]), int(r[8]), int(r[9]), r[11])
                match = [d for d in doms if d[:8] == want]
                assert match, (want, doms)
                assert abs(match[0][8] - float(r[13])) <= 0.051 and abs(match[0][9] - float(r[14])) <= 0.051, (want, match[0], r[13], r[14])
                found += 1
        else:
            best = max(doms, key=lambda d: d[8])
            assert best[7] == "-" and best[8] > 50.0
            found += 1
    assert found >= (len(rows) if rows else 1)


def test_a_dozen_nodes_against_the_oracle_restatement(oracle):
    """A 12-node nucleotide model with 40-residue windows (one-residue alignments, envelopes at the target's first residues,
    windows of the minimum length): stage counts, windows, scores and coordinates as in the tests above."""
    import lt_oracle_check as lc
    hmm = _cut_model(load_hmms("bmyD")[0], 100, 112, "bmyD_12")
    pli = plan7.LongTargetsPipeline(hmm.alphabet, block_length=16384, E=1000.0, window_length=40)
    seq = lc.synthetic_chromosome(hmm, 40_000, seed=100, block_length=pli.block_length, max_length=40, long_copies=False)
    block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chrD", sequence=seq)])
    hits = host_pipeline.host_nhmmer(oracle, hmm, block, pipeline=pli)
    windows = lc.oracle_windows(pli, hmm, seq)
    lc.check_hits_against_oracle(pli, hmm, seq, hits, oracle=windows)
    assert lc.check_hit_coordinates_against_oracle(pli, hmm, seq, hits, oracle=windows)[0] >= len(hits) - 2 and len(hits) >= 10


def test_per_query_state_is_settled_before_a_search_is_handed_to_a_worker(libp7x):
    """hmmer.nhmmer keeps two searches in flight, and [hmm] * n puts the SAME object into both.  What a search
    reads from and writes to its query (the profile it scans with, the replacement of hmm.max_length, plan7.pyx:7336-7354)
    is settled by LongTargetsPipeline._prepare_query on the caller's thread, in query order: the first preparation scans with
    the file's MAXL, every later one with the replaced value, whatever the timing of the searches themselves."""
    hmm = load_hmms("bmyD")[0]
    block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="t", sequence=np.zeros(5000, dtype=np.uint8))])
    pli = plan7.LongTargetsPipeline(hmm.alphabet, window_beta=1e-3)
    om1, cfg1 = pli._prepare_query(hmm, block)
    om2, cfg2 = pli._prepare_query(hmm, block)
    assert om1._info.max_length == 1736 and om2._info.max_length == cfg1.evalue_window_length == cfg2.evalue_window_length == hmm.max_length
    assert 1203 < hmm.max_length < 1736
    # the residency token of a packed image is minted once, under a lock
    import threading
    pk = block.packed()
    toks = []
    def grab():
        with plan7._RESIDENT_LOCK:
            tok = getattr(pk, "_resident_token", None)
            if tok is None:
                tok = next(plan7._RESIDENT_TOKENS); pk._resident_token = tok
        toks.append(tok)
    ts = [threading.Thread(target=grab) for _ in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert len(set(toks)) == 1
