"""Stochastic traceback ensembles of multi-domain regions on the device (p7x_ensemble.hip: multihit Forward fill with
the integer thresholds of every traceback choice, then one wavefront per region walking the region's 200 samples)
against the host twin (p7x_domaindef.cpp), which is pinned to the golden domain tables by tests/test_host_domaindef.py.

The sampled domains (sample number, residue and node coordinates, in order) and the per-residue sums of the sampled
null2 odds ratios must be IDENTICAL -- integer for integer, float bit for float bit: both sides form the Forward matrix with
the same operations in the same order and take every choice through the same thresholds (p7x_choice.hpp) from the same
generator stream (reference: p7_domaindef.pxd:23-59, p7_spensemble.pxd:3-39, re-seeding plan7.pyx:5684-5688)."""
import numpy as np
import pytest

import bench
from conftest import load_hmms, random_hmm
from pyhmmer_amd import plan7
from test_gpu_envelopes import _records, _repeat_protein
from test_gpu_filters import _model_block
from pyhmmer_amd import easel

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _host_twin_in_the_device_order():
    """This module validates the KERNELS against the host twin of the same arithmetic: option "host_order" = 1 makes the host
    code sum in the device's lane-chunk order (the product's default is upstream's striped order, which the device reaches
    through its near-tie guards: tests/test_gpu_oracle_domains.py)."""
    from pyhmmer_amd import _lib
    _lib.set_debug_option("host_order", 1)
    yield
    _lib.set_debug_option("host_order", -1)


def _same_ensemble(db, om, target, start, end, seed=42):
    sd, dd, nd = db.ensemble(om, target, start, end, seed=seed, device=True)
    sh, dh, nh = db.ensemble(om, target, start, end, seed=seed, device=False)
    assert sh == 0
    assert sd == 0, (target, start, end, sd)
    assert dd.shape == dh.shape, (target, start, end, dd.shape, dh.shape, dd[:4], dh[:4])
    if not np.array_equal(dd, dh):
        bad = int(np.argmax((dd != dh).any(axis=1)))
        raise AssertionError((target, start, end, bad, dd[bad], dh[bad]))
    bits_d, bits_h = nd.view(np.uint32), nh.view(np.uint32)
    if not np.array_equal(bits_d, bits_h):
        bad = int(np.argmax(bits_d != bits_h))
        raise AssertionError((target, start, end, "null2 sums", bad, float(nd[bad]), float(nh[bad]), int((bits_d != bits_h).sum())))
    return len(dd)


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam", "KR", "LuxC"])
def test_device_ensembles_equal_host_ensembles_on_fixture_hits(name, models, proteome):
    """Every hit of the fixture proteome: the whole target as one region, and its halves (regions that cut through domains)."""
    db = plan7.SequenceDatabase(proteome)
    index = {s.name: i for i, s in enumerate(proteome)}
    total = 0
    for hmm in models[name][:6]:
        bg = plan7.Background(hmm.alphabet)
        om = plan7.OptimizedProfile(hmm, bg, 400)
        hits = plan7.Pipeline(hmm.alphabet, E=1e3, domE=1e3).search_hmm(hmm, db)
        for h in list(hits)[:12]:
            t = index[h.name]
            L = len(proteome[t])
            total += _same_ensemble(db, om, t, 1, L)
            if L >= 40:
                total += _same_ensemble(db, om, t, 1, L // 2)
                total += _same_ensemble(db, om, t, L // 3, L)
    assert total > 0


def test_device_ensembles_of_repeat_proteins_and_other_seeds():
    """Targets with 2 to 12 copies of the domain (many domains per sample), several seeds."""
    hmm = load_hmms("KR")[0]
    abc = hmm.alphabet
    seqs = [easel.DigitalSequence(abc, name=f"rep{n}", sequence=_repeat_protein(hmm, n, 15 * n, 50 + n)) for n in (2, 3, 5, 8, 12)]
    db = plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, seqs))
    om = plan7.OptimizedProfile(hmm, plan7.Background(abc), 400)
    for t, s in enumerate(seqs):
        for seed in (42, 1, 7, 123456789):
            n = _same_ensemble(db, om, t, 1, len(s), seed=seed)
            assert n >= 200                        # at least one domain per sample


@pytest.mark.parametrize("M", [5, 64, 65, 150, 256, 300, 384, 478, 500, 640, 768, 1000, 1024, 1100, 1500, 2048, 2049, 3000, 5000, 8192])
def test_device_ensembles_for_every_kernel_instantiation(M):
    """Random models, one per nodes-per-lane instantiation of the Forward fill."""
    hmm = random_hmm(M, seed=3000 + M)
    nhom = 3 if M <= 2048 else 1
    blk = _model_block(hmm, 100, nhom, seed=M)          # the homologs (fragments of sequences emitted by the model) come last
    db = plan7.SequenceDatabase(blk)
    om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 400)
    n = len(blk)
    for t in range(n - nhom, n):
        _same_ensemble(db, om, t, 1, len(blk[t]))
    _same_ensemble(db, om, 0, 1, len(blk[0]))            # and a background target: a region without a domain to speak of


def test_search_with_device_ensembles_equals_search_with_host_ensembles():
    """The whole search, config-2 shaped with planted domains (pairs of them in some targets: multi-domain regions): every
    field of every hit the same whether the ensembles are sampled on the device or by the host workers."""
    hmm = load_hmms("KR")[0]
    abc = hmm.alphabet
    flat, off, ln, planted = bench.make_workload(hmm, 20_000, 300, 11, planted_frac=0.03)
    db = plan7.SequenceDatabase.from_packed(abc, flat, off, ln)
    dev = plan7.Pipeline(abc).search_hmm(hmm, db)
    host = plan7.Pipeline(abc, host_ensembles=True).search_hmm(hmm, db)
    assert sum(h.nclustered for h in dev) > 0
    assert [(h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) for h in dev] == [(h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) for h in host]
    assert _records(dev) == _records(host)
    # repeat proteins: regions with many domains each
    seqs = [easel.DigitalSequence(abc, name=f"rep{n}", sequence=_repeat_protein(hmm, n, 1 + n % 3, 70 + n)) for n in range(2, 14)]
    db2 = plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, seqs))
    dev2 = plan7.Pipeline(abc).search_hmm(hmm, db2)
    host2 = plan7.Pipeline(abc, host_ensembles=True).search_hmm(hmm, db2)
    assert _records(dev2) == _records(host2)
    assert [(h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) for h in dev2] == [(h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) for h in host2]


def test_without_the_near_tie_guard_device_and_host_twin_still_agree():
    """The host twin sums in the device's order (lane chunks, the wavefront's scan and reduction trees), so the optimal-accuracy
    alignments agree without the guard that used to send near-ties back to the host: every integer field of every domain."""
    hmm = load_hmms("KR")[0]
    abc = hmm.alphabet
    flat, off, ln, planted = bench.make_workload(hmm, 50_000, 300, 7, planted_frac=0.02)
    db = plan7.SequenceDatabase.from_packed(abc, flat, off, ln)
    dev = plan7.Pipeline(abc, oa_guard=0.0).search_hmm(hmm, db)
    assert dev.guard_counts["oa_redone"] == 0
    host = plan7.Pipeline(abc, host_envelopes=True, host_regions=True).search_hmm(hmm, db)
    a, b = sorted(_records(dev), key=lambda r: r[0]), sorted(_records(host), key=lambda r: r[0])
    assert [r[0] for r in a] == [r[0] for r in b]
    ndiff = ndom = 0
    for (name, _, da), (_, _, dbb) in zip(a, b):
        assert len(da) == len(dbb), name
        for (ia, _), (ib, _) in zip(da, dbb):
            ndom += 1
            ndiff += ia != ib
    assert ndom >= 900
    assert ndiff == 0, (ndiff, ndom)
