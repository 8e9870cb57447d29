"""Device rescoring of domain envelopes (p7x_envelope.hip: Forward + Backward + decoding + null2 + optimal
accuracy + traceback in one kernel) against the host implementation of the same steps (p7x_domaindef.cpp,
itself pinned to the golden domain tables by tests/test_host_domaindef.py).

Integer outputs (envelope / alignment / model coordinates, alignment strings, posterior-probability line) must be
identical for every domain (float near-ties of the optimal-accuracy traceback are detected on the device and repeated by
the host twin, p7x_pipeline_cfg.oa_guard); scores agree to ENV_TOL_BITS (float32 sums in a different association order)."""
import numpy as np
import pytest

import bench
from conftest import load_hmms, random_hmm, synthetic_block
from pyhmmer_amd import easel, hmmer, plan7
from test_gpu_filters import _model_block

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

ENV_TOL_BITS = 5e-3
ENV_TOL_REL_LONG = 5e-5  # models of thousands of nodes (scores and null2 corrections of thousands / hundreds of bits, each a float32
                         # sum over thousands of terms): relative, on top of ENV_TOL_BITS


def _records(hits):
    out = []
    for h in hits:
        doms = []
        for d in h.domains:
            a = d.alignment
            doms.append(((d.env_from, d.env_to, a.target_from, a.target_to, a.hmm_from, a.hmm_to, a.target_sequence,
                          a.hmm_sequence, a.identity_sequence, a.posterior_probabilities),
                         (d.score, d.bias, d.accuracy * 10.0)))
        out.append((h.name, (h.score, h.bias), doms))
    return out


def _compare(hmm, db, rtol=1e-5, **opts):
    dev = _records(plan7.Pipeline(hmm.alphabet, **opts).search_hmm(hmm, db))
    host = _records(plan7.Pipeline(hmm.alphabet, host_envelopes=True, host_regions=True, **opts).search_hmm(hmm, db))
    # the region scan alone (same envelope kernel on both sides): everything must be identical, bit for bit
    a = plan7.Pipeline(hmm.alphabet, host_regions=True, **opts).search_hmm(hmm, db)
    assert _records(a) == dev
    assert [(h.nregions, h.nclustered, h.nenvelopes, h.nexpected) for h in a] == \
           [(h.nregions, h.nclustered, h.nenvelopes, h.nexpected) for h in plan7.Pipeline(hmm.alphabet, **opts).search_hmm(hmm, db)]
    # hits whose scores differ by less than the tolerance may swap places in the ranking: compare by name
    dev, host = sorted(dev, key=lambda r: r[0]), sorted(host, key=lambda r: r[0])
    assert [r[0] for r in dev] == [r[0] for r in host]
    ndom = 0
    for (name, sa, da), (_, sb, dbb) in zip(dev, host):
        assert np.allclose(sa, sb, atol=ENV_TOL_BITS, rtol=rtol), name
        assert len(da) == len(dbb), name
        for (ia, fa), (ib, fb) in zip(da, dbb):
            assert np.allclose(fa, fb, atol=ENV_TOL_BITS, rtol=rtol), (name, fa, fb)
            ndom += 1
            # The optimal-accuracy alignment is an argmax over float32 sums.  The device kernel flags every choice on its
            # trace that lies within the guard band of the runner-up (cfg.oa_guard) and the host twin repeats those
            # envelopes: coordinates, alignment strings and the posterior line are identical for EVERY domain.
            assert ia == ib, (name, ia, ib)
    return len(dev), ndom


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam", "KR", "LuxC"])
def test_device_envelopes_equal_host_envelopes_on_fixtures(name, models, proteome):
    db = plan7.SequenceDatabase(proteome)
    total = 0
    for hmm in models[name]:
        total += _compare(hmm, db, E=1e3, domE=1e3)[1]
    assert total > 0


def test_device_envelopes_on_planted_workload():
    """BASELINE config-2 shape with 2 % planted domains: ~1,000 envelopes through the kernel in one batch."""
    hmm = load_hmms("KR")[0]
    flat, off, ln, planted = bench.make_workload(hmm, 50_000, 300, 7, planted_frac=0.02)
    db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, off, ln)
    nhits, ndom = _compare(hmm, db)
    assert nhits >= 900 and ndom >= nhits
    # without the near-tie guard nothing is repeated ...
    hits = plan7.Pipeline(hmm.alphabet, oa_guard=0.0).search_hmm(hmm, db)
    assert hits.guard_counts["oa_redone"] == 0
    # ... with it (the default: 4e-6) a small fraction of the envelopes is flagged and repeated by the host code -- here, with
    # the twin in the device's order, to the same result
    guarded = plan7.Pipeline(hmm.alphabet).search_hmm(hmm, db)
    redone = guarded.guard_counts["oa_redone"]
    assert 0 < redone <= ndom // 20, (redone, ndom)
    assert _records(guarded) == _records(hits)


@pytest.mark.parametrize("M", [5, 64, 65, 150, 256, 300, 384, 478, 500, 640, 768, 1000, 1024, 1100, 1500, 2048, 2049, 3000, 5000, 8192])
def test_device_envelopes_for_every_kernel_instantiation(M):
    """Random models, one per nodes-per-lane instantiation of the envelope kernel."""
    hmm = random_hmm(M, seed=3000 + M)
    blk = _model_block(hmm, 300, 40, seed=M)
    db = plan7.SequenceDatabase(blk)
    nhits, ndom = _compare(hmm, db, E=1e3, domE=1e3, rtol=ENV_TOL_REL_LONG if M > 2048 else 1e-5)
    assert ndom > 0 or M < 64


@pytest.mark.parametrize("M", [1100, 2048, 2049, 3000, 5000])
def test_long_models_search_end_to_end(M):
    """M > 1024: every stage on the device, the emission tables of the parsers and the envelope kernel read through L2;
    M > 2048 (the reference has no model-length limit, plan7.pyx:6156-6262): the long-model instantiations with the
    lane's row state in scratch memory, M > 4096 with the transition tables through L2 as well."""
    hmm = random_hmm(M, seed=4000 + M)
    blk = _model_block(hmm, 150, 12, seed=M)
    db = plan7.SequenceDatabase(blk)
    nhits, ndom = _compare(hmm, db, E=1e3, domE=1e3, rtol=ENV_TOL_REL_LONG if M > 2048 else 1e-5)
    assert ndom >= 12


def _repeat_protein(hmm, nrep, spacer, seed):
    """A long target: <nrep> copies of a sequence emitted by the model, separated by random spacers."""
    rng = np.random.default_rng(seed)
    t = hmm.transition_probabilities.astype(np.float64)
    mat, ins = hmm.match_emissions.astype(np.float64), hmm.insert_emissions.astype(np.float64)
    cmat = np.cumsum(mat / np.maximum(mat.sum(axis=1, keepdims=True), 1e-30), axis=1)
    cins = np.cumsum(ins / np.maximum(ins.sum(axis=1, keepdims=True), 1e-30), axis=1)
    ct = np.zeros((hmm.M + 1, 4))
    s3 = np.maximum(t[:, 0:3].sum(axis=1), 1e-30)
    ct[:, 0], ct[:, 1] = t[:, 0] / s3, (t[:, 0] + t[:, 1]) / s3
    ct[:, 2] = t[:, 3] / np.maximum(t[:, 3] + t[:, 4], 1e-30)
    ct[:, 3] = t[:, 5] / np.maximum(t[:, 5] + t[:, 6], 1e-30)
    parts = []
    for _ in range(nrep):
        parts.append(bench.emit_from_model(hmm, rng, (cmat, cins, ct)))
        parts.append(rng.integers(0, 20, size=spacer).astype(np.uint8))
    return np.concatenate(parts)


def test_very_long_targets_and_region_overflow():
    """A 30,000-residue target with 60 domains (envelope slabs sized for long envelopes, many regions) and a target
    with more regions than the device scan keeps (>128: the whole block falls back to the host scan); both must
    give the host twin's answer."""
    hmm = load_hmms("KR")[0]
    abc = hmm.alphabet
    seqs = [easel.DigitalSequence(abc, name="long60", sequence=_repeat_protein(hmm, 60, 250, 1)[:30000]),
            easel.DigitalSequence(abc, name="tight4", sequence=_repeat_protein(hmm, 4, 3, 2))]
    seqs += list(synthetic_block(200, 300, seed=3, alphabet=abc))
    db = plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, seqs))
    nhits, ndom = _compare(hmm, db)
    assert nhits >= 2 and ndom >= 60
    many = easel.DigitalSequence(abc, name="many150", sequence=_repeat_protein(hmm, 150, 120, 4))
    assert len(many) <= 100000
    db2 = plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, [many] + seqs[2:50]))
    hits = plan7.Pipeline(abc).search_hmm(hmm, db2)
    assert hits[0].name == "many150" and hits[0].nregions > 128 and len(hits[0].domains) >= 140
    nhits2, ndom2 = _compare(hmm, db2)
    assert ndom2 >= 140


def test_more_survivors_than_the_row_buffers_were_sized_for():
    """6,000 targets that are all homologs: more Forward survivors than the first sizing of the row buffers
    (max(4096, N/64)), so the cascade tail is repeated with larger buffers; every target must come back as a hit, and
    equal to the host twin."""
    hmm = load_hmms("PF02826")[0]
    abc = hmm.alphabet
    rng = np.random.default_rng(11)
    seqs = []
    for t in range(6000):
        dom = _repeat_protein(hmm, 1, int(rng.integers(0, 30)), 100 + t)
        seqs.append(easel.DigitalSequence(abc, name=f"h{t}", sequence=dom))
    db = plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, seqs))
    hits = plan7.Pipeline(abc).search_hmm(hmm, db)
    assert hits.stage_counts["fwd"] > 4096 and len(hits) == hits.stage_counts["fwd"]
    assert sorted(h.name for h in hits)[:3] == ["h0", "h1", "h10"]
    nhits, ndom = _compare(hmm, db)
    assert nhits == len(hits)


def test_degenerate_databases(models):
    hmm = models["PF02826"][0]
    abc = hmm.alphabet
    pli = plan7.Pipeline(abc)
    empty = pli.search_hmm(hmm, easel.DigitalSequenceBlock(abc, []))
    assert len(empty) == 0 and empty.searched_sequences == 0
    tiny = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name=f"s{L}", sequence=np.full(L, L % 20, dtype=np.uint8))
                                            for L in (1, 2, 3, 5, 8, 0, 1)])
    res = pli.search_hmm(hmm, tiny)
    assert len(res) == 0 and res.searched_sequences == 7 and res.searched_residues == 20
    assert [len(h) for h in hmmer.hmmscan(tiny, [hmm])] == [0] * 7
