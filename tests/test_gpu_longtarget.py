"""GPU tests of the long-target (nhmmer) path: the device SSV scan against the oracle's sequential
p7_SSVFilter_longtarget, and LongTargetsPipeline / hmmer.nhmmer end to end against the reference's nhmmer fixtures
(reference tests/test_hmmer.py:631-795)."""
import ctypes as C

import numpy as np
import pytest

import host_pipeline
from conftest import GOLDEN, golden_table, load_hmms
from pyhmmer_amd import _lib, easel, hmmer, plan7
from test_host_longtarget import _read, _rows, check_bmyd2_table, check_nhmmer_table

pytestmark = pytest.mark.gpu


def rows_agree(dev, ref):
    """Device search against the CPU harness: names, coordinates, strands and flags identical; scores, biases and E-values
    to float32 summation-order noise (the device rescoring the envelopes sums Forward / Backward in another order)."""
    assert len(dev) == len(ref), (len(dev), len(ref))
    for a, b in zip(dev, ref):
        assert a[:8] == b[:8], (a, b)                        # name, model and target coordinates, envelope, strand
        assert a[11:] == b[11:], (a, b)                      # reported / included
        assert a[8] == pytest.approx(b[8], rel=2e-3, abs=1e-300), (a, b)
        assert a[9] == pytest.approx(b[9], abs=3e-3) and a[10] == pytest.approx(b[10], abs=3e-3), (a, b)


# binary16 cells (the default since round 6: v_pk_add_f16 clamp + v_pk_maximum3_f16) and the int16 flavour behind the seam
SSV_KERNELS = {"the library's choice": -1, "row maximum in every row": 3, "row maximum every second row": 4,
               "int16 cells, the library's choice of rows": 5, "int16 cells, row maximum every second row": 7}


@pytest.fixture
def ssv_kernel():
    """Selects the long-target SSV kernel through the test seam; the library's own choice again afterwards."""
    yield lambda variant: _lib.set_debug_option("ssv_kernel", variant)
    _lib.set_debug_option("ssv_kernel", -1)


def device_seeds(om, cfg, residues, complement, cap=1 << 16):
    seeds = np.zeros((cap, 3), dtype=np.int64)
    d = np.ascontiguousarray(residues, dtype=np.uint8)
    n = _lib.lib().p7x_ssv_longtarget_seeds(C.byref(cfg), om._handle, 0, d.ctypes.data, len(d), int(complement), seeds.ctypes.data, cap)
    assert 0 <= n <= cap, _lib.last_error()
    return seeds[:n]


@pytest.mark.parametrize("model,target", [("bmyD", "BGC0001090.gbk"), ("bmyD", "1390.SAMEA104415756.OFHT01000022.fna"),
                                          ("RF00001", "1390.SAMEA104415756.OFHT01000024.fna")])
def test_device_ssv_seeds_equal_the_sequential_scan(oracle, model, target, ssv_kernel):
    """The window seeds (first residue, last model node, diagonal length) that come out of the device scan + the host's
    sequential bookkeeping equal those of the oracle's p7_SSVFilter_longtarget, on both strands of every fixture target
    (whole target as one block), with every kernel."""
    hmm = load_hmms(model)[0]
    seqs = _read(target, hmm.alphabet)
    pli = plan7.LongTargetsPipeline(hmm.alphabet, block_length=1 << 30)
    bg = pli.background
    om = plan7.OptimizedProfile(hmm, bg, 400)
    op = oracle.OracleProfile(hmm, bg, 400)
    cfg = pli._cfg()
    seq = np.asarray(seqs[0].sequence, dtype=np.uint8)
    total = 0
    for strand in (0, 1):
        blk = seq if strand == 0 else host_pipeline.DNA_COMP[seq[::-1]]
        want = oracle.ssv_longtarget(op, blk, hmm.max_length, pli.F1)
        for what, variant in SSV_KERNELS.items():
            ssv_kernel(variant)
            got = device_seeds(om, cfg, seq, strand)
            assert got.tolist() == want.tolist(), (model, target, strand, what)
        total += len(want)
    assert total > 0


def test_device_ssv_on_a_synthetic_chromosome(oracle, ssv_kernel):
    """2 Mbp of i.i.d. ACGT with 40 planted copies of model segments (some adjacent, some overlapping a chunk boundary of
    the device scan): seeds equal the sequential scan; a few N runs exercise the degenerate-residue path."""
    hmm = load_hmms("bmyD")[0]
    rng = np.random.default_rng(7)
    L = 2_000_000
    seq = rng.integers(0, 4, size=L).astype(np.uint8)
    cons = np.argmax(hmm.match_emissions[1:], axis=1).astype(np.uint8)
    for c in range(40):
        a = int(rng.integers(0, hmm.M - 120)); n = int(rng.integers(60, 600)); n = min(n, hmm.M - a)
        pos = int(rng.integers(0, L - n))
        seg = cons[a:a + n].copy()
        mut = rng.random(n) < 0.15
        seg[mut] = rng.integers(0, 4, size=int(mut.sum()))
        seq[pos:pos + n] = seg if c % 2 == 0 else host_pipeline.DNA_COMP[seg[::-1]]
    for s in (1000, 777_000, 1_999_000):
        seq[s:s + 50] = 15          # N
    pli = plan7.LongTargetsPipeline(hmm.alphabet, block_length=1 << 30)
    om = plan7.OptimizedProfile(hmm, pli.background, 400)
    op = oracle.OracleProfile(hmm, pli.background, 400)
    for strand in (0, 1):
        blk = seq if strand == 0 else host_pipeline.DNA_COMP[seq[::-1]]
        want = oracle.ssv_longtarget(op, blk, hmm.max_length, pli.F1)
        assert len(want) > 20
        for what, variant in SSV_KERNELS.items():
            ssv_kernel(variant)
            got = device_seeds(om, pli._cfg(), seq, strand)
            assert got.tolist() == want.tolist(), (strand, what)


def test_nhmmer_bmyd_tables_through_the_device(oracle):
    """hmmer.nhmmer == the CPU harness (oracle scan + host tail) on the bmyD fixtures, and the golden tables."""
    hmm = load_hmms("bmyD")[0]
    for target, table in (("BGC0001090.gbk", "bmyD1.tbl"), ("1390.SAMEA104415756.OFHT01000022.fna", "bmyD2.tbl")):
        seqs = _read(target, hmm.alphabet)
        hits = next(hmmer.nhmmer(hmm, seqs))
        ref = host_pipeline.host_nhmmer(oracle, hmm, seqs)
        rows_agree(_rows(hits), _rows(ref))
        # the envelopes through the envelope kernel's long-target instantiation (host_envelopes=2: always; by default a
        # handful of envelopes stays with the host workers) and through the host workers: the same rows, the same table
        for where in (1, 2):
            forced = next(hmmer.nhmmer(hmm, seqs, host_envelopes=where))
            rows_agree(_rows(forced), _rows(ref))
            (check_bmyd2_table if table == "bmyD2.tbl" else check_nhmmer_table)(forced, golden_table(table))
        if table == "bmyD2.tbl":
            check_bmyd2_table(hits, golden_table(table))
        else:
            check_nhmmer_table(hits, golden_table(table))
        assert hits.searched_residues == 2 * len(seqs[0]) and hits.searched_sequences == 1
    # from a file object, one strand
    with easel.SequenceFile(GOLDEN / "seqs" / "BGC0001090.gbk", digital=True, alphabet=hmm.alphabet) as f:
        one = list(hmmer.nhmmer(hmm, f, strand="crick"))[0]
    assert [h.best_domain.strand for h in one.reported] == ["-"] and one.strand == "crick"
    assert next(hmmer.nhmmer([], seqs), None) is None                     # reference test_no_queries


def test_nhmmer_rf00001_known_answers():
    """reference test_rf0001_genome_file / _wlen_3878 through the device path."""
    hmm = load_hmms("RF00001")[0]
    with easel.SequenceFile(GOLDEN / "seqs" / "1390.SAMEA104415756.OFHT01000024.fna", digital=True, alphabet=hmm.alphabet) as f:
        hits = list(hmmer.nhmmer(hmm, f))[0]
    assert len(hits) == 1 and float("%.2g" % hits[0].evalue) == pytest.approx(2.5e-17, rel=1e-6) and hits[0].best_domain.strand == "-"
    seqs = _read("1390.SAMEA104415756.OFHT01000024.fna", hmm.alphabet)
    hits = list(hmmer.nhmmer(hmm, seqs, window_length=3878))[0]
    assert len(hits) == 2 and float("%.2g" % hits[0].evalue) == pytest.approx(5.4e-17, rel=1e-6) and hits[1].evalue == pytest.approx(0.3, abs=0.005)


def test_nhmmer_dealt_over_devices_equals_one_device():
    """hmmer.nhmmer(devices=[...]): the (target, block, strand) units of one search dealt over several devices (here the
    same device three and five times: every part scans only the chunks of its own blocks, cfg.lt_part / lt_nparts) and
    finished together == the one-device search, on a 6 Mbp synthetic chromosome (24 blocks x 2 strands, planted hits
    next to block boundaries) plus a short second target, and on the golden bmyD2 table."""
    import bench_workloads as bw
    hmm = load_hmms("bmyD")[0]
    abc = hmm.alphabet
    big = bw.make_chromosome(hmm, 6_000_000, planted=60, seed=11)
    W, Cv = 0x40000, hmm.max_length
    cons = np.argmax(hmm.match_emissions[1:], axis=1).astype(np.uint8)
    for b in (3, 7, 11):                     # a copy of the model's first 900 nodes across the seam of blocks b-1 / b
        at = b * (W - Cv) + Cv // 2 - 450
        big[at:at + 900] = cons[:900]
    small = bw.make_chromosome(hmm, 150_000, planted=3, seed=12)
    block = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name="chrA", sequence=big),
                                             easel.DigitalSequence(abc, name="ctgB", sequence=small)])
    # host_envelopes=1: where the envelopes are rescored (host workers or envelope kernel) is decided by their number, which a
    # part sees less of than the whole; pinned to the host workers the dealt search equals the whole one bit for bit
    one = next(hmmer.nhmmer(hmm, block, host_envelopes=1))
    assert len(one) > 40 and any(h.duplicate for h in one)
    rows_agree(_rows(next(hmmer.nhmmer(hmm, block, devices=[0, 0]))), _rows(one))
    for devs in ([0, 0, 0], [0] * 5):
        many = next(hmmer.nhmmer(hmm, block, devices=devs, host_envelopes=1))
        assert _rows(many) == _rows(one), devs
        assert [(h.evalue, h.reported, h.included, h.duplicate) for h in many] == [(h.evalue, h.reported, h.included, h.duplicate) for h in one]
        assert many.stage_counts == one.stage_counts and many.searched_residues == one.searched_residues
    seqs = _read("1390.SAMEA104415756.OFHT01000022.fna", abc)
    check_bmyd2_table(next(hmmer.nhmmer(hmm, seqs, devices=[0, 0])), golden_table("bmyD2.tbl"))


@pytest.mark.parametrize("M", [60, 150, 250, 330, 380, 440, 500, 560, 630, 700, 760, 880, 1000, 1270, 1500, 2040, 2500, 3060, 3500, 5000])
def test_device_ssv_every_register_count(M, oracle, ssv_kernel):
    """One model length per instantiation (registers per lane) of the long-target SSV kernel, with the row maximum in every
    second row (lowered threshold, virtual node M + 1, exact repeat; the library's choice for models whose cells lose at
    most 16 score units per row -- not these, whose sharpened emissions lose 60-110) and in every row, on a 400 kb random
    sequence with planted stretches of the model's consensus and a run of N: window seeds equal the oracle's sequential
    p7_SSVFilter_longtarget, both strands."""
    abc = easel.Alphabet.dna()
    from conftest import random_hmm
    hmm = random_hmm(M, seed=9000 + M, alphabet=abc)
    rng = np.random.default_rng(M)
    L = 400_000
    seq = rng.integers(0, 4, size=L).astype(np.uint8)
    cons = np.argmax(hmm.match_emissions[1:], axis=1).astype(np.uint8)
    for c in range(30):
        a = int(rng.integers(0, max(1, M - 40)))
        n = min(int(rng.integers(30, 400)), M - a)
        pos = int(rng.integers(0, L - n))
        seg = cons[a:a + n].copy()
        mut = rng.random(n) < 0.1
        seg[mut] = rng.integers(0, 4, size=int(mut.sum()))
        seq[pos:pos + n] = seg if c % 2 == 0 else host_pipeline.DNA_COMP[seg[::-1]]
    seq[5000:5030] = 15                                   # a run of N: the degenerate-residue path
    pli = plan7.LongTargetsPipeline(abc, block_length=1 << 30)
    om = plan7.OptimizedProfile(hmm, pli.background, 400)
    op = oracle.OracleProfile(hmm, pli.background, 400)
    total = 0
    for strand in (0, 1):
        blk = seq if strand == 0 else host_pipeline.DNA_COMP[seq[::-1]]
        want = oracle.ssv_longtarget(op, blk, hmm.max_length, pli.F1)
        for what, variant in SSV_KERNELS.items():
            ssv_kernel(variant)
            got = device_seeds(om, pli._cfg(), seq, strand)
            assert got.tolist() == want.tolist(), (M, strand, what)
        total += len(want)
    assert total >= 10, total


def test_long_target_envelopes_on_the_device_equal_the_host_workers():
    """A hit-rich search (250 planted copies on 3 Mbp: the case the device path is for): the envelope kernel's long-target
    instantiation (two rounds: the envelope, then the envelope trimmed to its alignment; composition-adjusted odds per
    envelope; Forward with the unmodified odds for the bias) against the host workers' rescore_isolated_domain: the same
    hits, coordinates and flags, scores to float32 noise."""
    import bench_workloads as bw
    hmm = load_hmms("bmyD")[0]
    abc = hmm.alphabet
    seq = bw.make_chromosome(hmm, 3_000_000, planted=250, seed=21)
    seq[1_000_000:1_000_040] = 15                                   # N inside a window
    block = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name="chrC", sequence=seq)])
    host = next(hmmer.nhmmer(hmm, block, host_envelopes=1))
    dev = next(hmmer.nhmmer(hmm, block, host_envelopes=2))
    assert len(host) > 200
    rows_agree(_rows(dev), _rows(host))
    assert dev.stage_counts == host.stage_counts
    auto = next(hmmer.nhmmer(hmm, block))                          # the default picks one of the two
    rows_agree(_rows(auto), _rows(host))


def test_contig_rich_target_set_is_one_scan(oracle):
    """An assembly-shaped target set -- 2 Mbp with 150 planted copies cut into ~500 contigs of 1 .. 12,000 residues, some
    shorter than the model, runs of N, a contig of N only -- goes through the device as ONE scan over the records laid end
    to end (the sentinel between two records ends every diagonal) and one device batch per stage; the hits equal the CPU
    harness that scans contig by contig with the oracle's sequential SSV filter."""
    import time
    import bench_workloads as bw
    hmm = load_hmms("bmyD")[0]
    abc = hmm.alphabet
    seq = bw.make_chromosome(hmm, 2_000_000, planted=150, seed=33)
    rng = np.random.default_rng(5)
    cuts, at = [], 0
    while at < len(seq):
        n = int(rng.choice([1, 40, 300, 900, 2500, 6000, 12000], p=[0.02, 0.05, 0.13, 0.2, 0.25, 0.2, 0.15]))
        cuts.append((at, min(len(seq), at + n)))
        at += n
    seqs = []
    for q, (a, b) in enumerate(cuts):
        s = seq[a:b].copy()
        if q % 37 == 5:
            s[len(s) // 3: len(s) // 3 + 25] = 15            # a run of N
        if q == 11:
            s[:] = 15
        seqs.append(easel.DigitalSequence(abc, name=f"ctg{q:05d}", sequence=s))
    assert len(seqs) > 400
    block = easel.DigitalSequenceBlock(abc, seqs)
    next(hmmer.nhmmer(hmm, block, host_envelopes=1))          # warm
    t0 = time.perf_counter()
    dev = next(hmmer.nhmmer(hmm, block, host_envelopes=1))
    dt = time.perf_counter() - t0
    ref = host_pipeline.host_nhmmer(oracle, hmm, seqs)
    assert len(ref) > 60
    rows_agree(_rows(dev), _rows(ref))
    assert dev.stage_counts == ref.stage_counts and dev.searched_residues == ref.searched_residues
    print(f"\n{len(seqs)} contigs, {len(seq)} residues: {dt * 1e3:.1f} ms per search, {len(dev)} hits")
    # the same search dealt over three parts
    many = next(hmmer.nhmmer(hmm, block, devices=[0, 0, 0], host_envelopes=1))
    assert _rows(many) == _rows(dev)


def test_targets_stay_on_the_device_between_searches(capfd):
    """cfg.lt_resident_key: the packed image of a block carries a token, and a later search of the same image -- by any
    query -- finds the targets on the device instead of uploading them again; a block that was changed has a new image,
    a new token and is uploaded; the results are those of a fresh block either way."""
    import bench_workloads as bw
    from pyhmmer_amd import _lib
    _lib.set_debug_option("trace_longtarget", 1)
    hmm = load_hmms("bmyD")[0]
    abc = hmm.alphabet
    seqs = [easel.DigitalSequence(abc, name=f"chr{i}", sequence=bw.make_chromosome(hmm, 400_000, planted=6, seed=70 + i)) for i in range(2)]
    block = easel.DigitalSequenceBlock(abc, seqs)
    pli = plan7.LongTargetsPipeline(abc)
    first = pli.search_hmm(hmm, block)
    e1 = capfd.readouterr().err
    second = plan7.LongTargetsPipeline(abc).search_hmm(hmm, block)           # another pipeline object, the same image
    e2 = capfd.readouterr().err
    assert "targets on the device: uploaded" in e1 and "targets on the device: resident" in e2
    assert _rows(first) == _rows(second) and len(first) >= 10
    extra = easel.DigitalSequence(abc, name="chr2", sequence=bw.make_chromosome(hmm, 300_000, planted=4, seed=90))
    block.append(extra)
    third = pli.search_hmm(hmm, block)
    e3 = capfd.readouterr().err
    assert "targets on the device: uploaded" in e3
    fresh = plan7.LongTargetsPipeline(abc).search_hmm(hmm, easel.DigitalSequenceBlock(abc, seqs + [extra]))
    assert _rows(third) == _rows(fresh) and len(third) > len(first)
    two = pli.search_hmm(hmm, block, devices=[0, 0])                        # parts of one search share the device's copy
    assert _rows(two) == _rows(third)
    _lib.set_debug_option("trace_longtarget", -1)


def test_a_target_buffer_with_gaps_is_packed_and_gives_the_same_hits():
    """The C-ABI takes any (dsq, offsets, lengths); only records that lie end to end with one sentinel between them are
    scanned where they are.  Here the records are laid out with gaps of ordinary residues between them and in a buffer
    that starts with junk: the library packs them first, and the hits are those of the contiguous layout."""
    import bench_workloads as bw
    hmm = load_hmms("bmyD")[0]
    abc = hmm.alphabet
    seqs = [easel.DigitalSequence(abc, name=f"c{i}", sequence=bw.make_chromosome(hmm, n, planted=p, seed=40 + i))
            for i, (n, p) in enumerate([(200_000, 3), (1_500, 0), (90_000, 2)])]
    block = easel.DigitalSequenceBlock(abc, seqs)
    pli = plan7.LongTargetsPipeline(abc)
    want = pli.search_hmm(hmm, block)
    assert len(want) >= 4
    cfg = pli._cfg()
    om = pli._windowed_om(hmm, 100000, cfg)
    gap = 1000
    total = sum(len(s) for s in seqs) + gap * (len(seqs) + 1)
    rng = np.random.default_rng(1)
    dsq = rng.integers(0, 4, size=total + 2, dtype=np.uint8)            # the gaps are residues, not sentinels
    offsets, lengths, pos = [], [], gap
    for s in seqs:
        dsq[pos:pos + len(s)] = s.sequence
        offsets.append(pos); lengths.append(len(s))
        pos += len(s) + gap
    offsets = np.array(offsets, dtype=np.int64); lengths = np.array(lengths, dtype=np.int64)
    names = (C.c_char_p * 3)(*[s.name.encode() for s in seqs])
    empty = (C.c_char_p * 3)(b"", b"", b"")
    out = C.c_void_p()
    st = _lib.lib().p7x_search_longtargets(C.byref(cfg), om._handle, 0, dsq.ctypes.data, offsets.ctypes.data, lengths.ctypes.data, 3,
                                           names, empty, empty, C.byref(out))
    assert st == 0, _lib.last_error()
    got = plan7.TopHits(hmm, out)
    assert _rows(got) == _rows(want)


def test_a_stream_of_queries_overlaps_searches_and_gives_the_same_hits():
    """hmmer.nhmmer keeps two searches in flight (the scan of one under the host tail of the other; the reference's
    hmmer/_nhmmer.py:24-56 runs its queries on worker threads): every query's hits equal those of the same search run alone,
    in query order, also when the queries differ (here: the fixture model and two sub-models cut out of it)."""
    import bench_workloads as bw
    full = load_hmms("bmyD")[0]
    abc = full.alphabet

    def cut(lo, hi, name):
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
        return h                     # no MAXL: every search computes the same window bound (an HMM WITH one has it replaced by
                                     # the first search, as in the reference, plan7.pyx:7346-7354 -- that would make 'alone' differ)

    hmms = [full, cut(100, 500, "bmyD_a"), cut(600, 1100, "bmyD_b")]
    seqs = [easel.DigitalSequence(abc, name=f"chr{i}", sequence=bw.make_chromosome(full, 600_000, planted=8, seed=170 + i)) for i in range(2)]
    block = easel.DigitalSequenceBlock(abc, seqs)
    order = [0, 1, 0, 2, 1, 0]
    alone = [_rows(plan7.LongTargetsPipeline(abc).search_hmm(q, block)) for q in hmms]
    want = [alone[i] for i in order]
    queries = [hmms[i] for i in order]
    assert len(alone[0]) >= 10 and len(alone[1]) >= 1 and len(alone[2]) >= 1
    assert [_rows(h) for h in hmmer.nhmmer(queries, block)] == want
    assert [_rows(h) for h in hmmer.nhmmer(queries, block, searches_in_flight=1)] == want
    assert [_rows(h) for h in hmmer.nhmmer(queries, block, searches_in_flight=3)] == want


def test_device_long_target_search_against_the_oracle_restatement():
    """hmmer.nhmmer (SSV scan, window filters, long-target Viterbi scan, Forward, Backward and region scan on the device,
    envelope kernel or host workers for the envelopes) on a 2 Mbp synthetic chromosome with more than a thousand SSV
    windows -- copies of the model on both strands, copies laid across the seams between blocks, hundreds of fragments
    shorter than 100 residues, low-complexity stretches -- against oracle/p7_oracle_lt.c, which restates the tail of
    p7_Pipeline_LongTarget apart from the product (plan7.pyx:7541-7664, p7_pipeline.pxd:131-143): the windows past every
    filter are counted alike, every hit lies inside a window the oracle lets through Forward, and its score and bias follow
    from its own envelope and alignment coordinates by the oracle's rule (1e-3 bit), the envelope's two Forward scores from
    its residues (5e-3 nat)."""
    import lt_oracle_check as lc
    hmm = load_hmms("bmyD")[0]
    pli = plan7.LongTargetsPipeline(hmm.alphabet)
    seq = lc.synthetic_chromosome(hmm, 2_000_000, seed=23, block_length=pli.block_length, max_length=hmm.max_length)
    block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name="chr2M", sequence=seq)])
    want = lc.oracle_windows(pli, hmm, seq)                     # once: a minute of scalar C
    hits = next(iter(hmmer.nhmmer([hmm], block)))
    nwin, nshort = lc.check_hits_against_oracle(pli, hmm, seq, hits, min_windows=1000, min_short=30, oracle=want)
    assert len(hits) >= 150
    # the same with the envelopes forced onto the envelope kernel, and onto the host workers
    for where in (1, 2):
        other = next(iter(hmmer.nhmmer([hmm], block, host_envelopes=where)))
        lc.check_hits_against_oracle(pli, hmm, seq, other, min_windows=1000, min_short=30, oracle=want)


def test_overlapping_searches_of_one_hmm_object_equal_the_sequential_ones():
    """hmmer.nhmmer overlaps consecutive searches (searches_in_flight = 2) on one shared pipeline; with [hmm] * n the same HMM
    object is in flight twice.  ADVICE r04: the per-query state is prepared in query order on the caller's thread, the scans
    of overlapping searches take turns at the device, the target set is uploaded once: every result equals the result of
    the queries run one after the other."""
    import bench_workloads as bw
    abc = load_hmms("bmyD")[0].alphabet
    chrom = bw.make_chromosome(load_hmms("bmyD")[0], 3_000_000, planted=25, seed=5)
    block = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name="chr", sequence=chrom)])
    hmm = load_hmms("bmyD")[0]
    seq = [_rows(h) for h in hmmer.nhmmer([hmm] * 5, block, searches_in_flight=1, window_beta=1e-3)]
    hmm = load_hmms("bmyD")[0]
    par = [_rows(h) for h in hmmer.nhmmer([hmm] * 5, block, searches_in_flight=3, window_beta=1e-3)]
    assert len(seq[0]) > 10 and seq == par
    assert seq[1] == seq[2] == seq[4]                      # from the second search on the replaced max_length is in force


def test_config4_at_full_size_seeds_against_the_oracle_and_the_hits_explained(oracle):
    """BASELINE configs[4] at its own size: bmyD against the bench's 250 Mbp chromosome.
    (1) The device's SSV seeds of the whole forward strand against the oracle's sequential p7_SSVFilter_longtarget on a
    sample of 262,144-residue blocks -- half of them around planted stretches, half background: every seed whose diagonal
    lies at least max_length inside a block must be in both lists (SSV has no J state: what a scan reports away from the
    ends of what it was given does not depend on where it was started).
    (2) The bench prints 110 hits for 50 planted stretches: the bmyD model scores its own reverse complement highly (a stretch
    planted on one strand is found on the other as well, E-values of 1e-30 ... 1e-150 next to 1e-50 ... 1e-270 on its own),
    so a stretch gives two hits, one per strand, now and then a third where the alignment breaks; half a dozen marginal
    hits (E > 0.1) lie on background.  Asserted: every hit with E < 1e-3 lies on a planted stretch, every stretch is found
    on its own strand, and nothing else explains a hit."""
    import bench_workloads as bw
    hmm = load_hmms("bmyD")[0]
    abc = hmm.alphabet
    seq, planted = bw.make_chromosome(hmm, 250_000_000, 50, return_planted=True)
    pli = plan7.LongTargetsPipeline(abc, block_length=1 << 30)
    om = plan7.OptimizedProfile(hmm, pli.background, 400)
    op = oracle.OracleProfile(hmm, pli.background, 400)
    got = device_seeds(om, pli._cfg(), seq, 0, cap=1 << 20)
    assert len(got) > 1000
    C_, W = int(hmm.max_length), 0x40000
    fwd = [p for p in planted if p[2] == 0]
    starts = [max(0, p[0] - W // 2) for p in fwd[:5]] + [10_000_000, 77_777_777, 123_456_789, 200_000_000, 249_000_000 - W]
    compared = 0
    for a in starts:
        blk = seq[a:a + W]
        want = oracle.ssv_longtarget(op, blk, hmm.max_length, pli.F1)
        inner = lambda s0, ln: s0 >= a + C_ and s0 + ln <= a + len(blk) - C_
        w = sorted((int(s[0]) + a, int(s[1]), int(s[2])) for s in want if inner(int(s[0]) + a, int(s[2])))
        g = sorted((int(s[0]), int(s[1]), int(s[2])) for s in got if inner(int(s[0]), int(s[2])))
        assert g == w, (a, len(g), len(w), [x for x in g if x not in w][:3], [x for x in w if x not in g][:3])
        compared += len(w)
    assert compared > 50
    block = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name="chrSyn", sequence=seq)])
    hits = next(hmmer.nhmmer(hmm, block))
    own = [0] * len(planted); other = [0] * len(planted); background = []
    for h in hits:
        dom = h.best_domain
        al = dom.alignment
        lo, hi = sorted((int(al.target_from), int(al.target_to)))
        strand = 0 if dom.strand == "+" else 1
        on = [i for i, (pos, n, st, a0) in enumerate(planted) if lo >= pos + 1 - 50 and hi <= pos + n + 50]
        if not on:
            background.append(h.evalue)
        elif planted[on[0]][2] == strand:
            own[on[0]] += 1
        else:
            other[on[0]] += 1
    assert all(e > 1e-3 for e in background) and len(background) <= 12, background
    assert all(c >= 1 for c in own), [(p, c) for p, c in zip(planted, own) if c == 0]                # every stretch, on its own strand
    assert sum(1 for c in other if c >= 1) >= 40                                                      # ... and nearly all on the other one
    assert max(own) <= 3 and max(other) <= 3
    assert len(hits) == sum(own) + sum(other) + len(background) and 95 <= len(hits) <= 125, (len(hits), sum(own), sum(other), len(background))
