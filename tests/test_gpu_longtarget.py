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


def device_seeds(om, cfg, residues, complement):
    cap = 1 << 16
    seeds = np.zeros((cap, 3), dtype=np.int64)
    d = np.ascontiguousarray(residues, dtype=np.uint8)
    n = _lib.lib().p7x_ssv_longtarget_seeds(C.byref(cfg), om._handle, 0, d.ctypes.data, len(d), int(complement), seeds.ctypes.data, cap)
    assert 0 <= n <= cap, _lib.last_error()
    return seeds[:n]


@pytest.mark.parametrize("model,target", [("bmyD", "BGC0001090.gbk"), ("bmyD", "1390.SAMEA104415756.OFHT01000022.fna"),
                                          ("RF00001", "1390.SAMEA104415756.OFHT01000024.fna")])
def test_device_ssv_seeds_equal_the_sequential_scan(oracle, model, target):
    """The window seeds (first residue, last model node, diagonal length) that come out of the device scan + the host's
    sequential bookkeeping equal those of the oracle's p7_SSVFilter_longtarget, on both strands of every fixture target
    (whole target as one block)."""
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
        got = device_seeds(om, cfg, seq, strand)
        assert got.tolist() == want.tolist(), (model, target, strand)
        total += len(want)
    assert total > 0


def test_device_ssv_on_a_synthetic_chromosome(oracle):
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
        got = device_seeds(om, pli._cfg(), seq, strand)
        assert len(want) > 20
        assert got.tolist() == want.tolist(), strand


def test_nhmmer_bmyd_tables_through_the_device(oracle):
    """hmmer.nhmmer == the CPU harness (oracle scan + host tail) on the bmyD fixtures, and the golden tables."""
    hmm = load_hmms("bmyD")[0]
    for target, table in (("BGC0001090.gbk", "bmyD1.tbl"), ("1390.SAMEA104415756.OFHT01000022.fna", "bmyD2.tbl")):
        seqs = _read(target, hmm.alphabet)
        hits = next(hmmer.nhmmer(hmm, seqs))
        ref = host_pipeline.host_nhmmer(oracle, hmm, seqs)
        assert _rows(hits) == _rows(ref)
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
