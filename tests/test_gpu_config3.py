"""BASELINE configs[3] at its own shape, against the oracle: Swiss-Prot-shaped targets (L ~ lognormal(5.65, 0.65) in
[30, 5000], half of them carrying a planted domain) x library profiles (M 20 ... 2000) that span every MSV kernel family
-- one target per lane (register tiles R), two and four lanes per target, the packed wave-per-target kernel for
M > 1021, and, through the hybrid split, the wave kernel on the longest target groups -- in ONE batch, i.e. through the
very launches hmmsearch issues for a Pfam-sized query stream (bench.py `pfam`).  Integer filter scores bit-exact and
per-profile stage counts equal to oracle.cascade_block; the sharded search ("devices=[0, 0]", the reference's
_ReverseSEARCHDispatcher, _hmmsearch.py:115-289) merged == whole, field by field (test_tophits.py:191-291)."""
import ctypes as C
import itertools
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np
import pytest

from pyhmmer_amd import _lib, easel, hmmer, plan7

pytestmark = pytest.mark.gpu

# oracle_lib.Record as a numpy record (C layout)
RECORD = np.dtype([("usc", "f4"), ("filtersc", "f4"), ("nullsc", "f4"), ("vfsc", "f4"), ("fwdsc", "f4"), ("P_msv", "f8"), ("P_bias", "f8"),
                   ("P_vit", "f8"), ("P_fwd", "f8"), ("xJ_msv", "i4"), ("xC_vit", "i4"), ("stage", "i4"), ("ran_vit", "i4")], align=True)
NTARGETS = 50_000
LIBRARY = 20_000
FIRST = 3_000          # the planted domains come from the first FIRST library entries; the test's profiles are among them
# model-length bins -> (profiles to take, what runs them)
BINS = [((20, 60), 4, "lane kernel, small register tiles"), ((61, 130), 6, "lane kernel"), ((131, 260), 6, "lane kernel"),
        ((261, 445), 6, "lane kernel, largest tiles"), ((446, 893), 6, "two lanes per target"),
        ((894, 1021), 3, "four lanes per target"), ((1022, 1500), 4, "packed wave kernel"), ((1501, 2000), 4, "packed wave kernel")]


@pytest.fixture(scope="module")
def workload():
    import bench_workloads as bw
    templates = bw.load_templates()
    lengths = bw.library_lengths(LIBRARY)
    cal = bw.Calibrator(templates[0].alphabet)
    picked = []
    for (lo, hi), want, _ in BINS:
        cand = [e for e in range(FIRST) if lo <= lengths[e] <= hi]
        picked.extend(cand[:want])
    assert len(picked) >= 36, "the library's first entries no longer cover every length bin"
    hmms = [cal.calibrate(bw.make_entry(templates, e, int(lengths[e]))) for e in picked]
    flat, offsets, lens, nplanted = bw.make_targets(NTARGETS, FIRST, templates, lengths, planted_frac=0.5)
    return hmms, flat, offsets, lens, nplanted


def test_workload_reaches_every_msv_family(workload):
    hmms = workload[0]
    Ms = sorted(h.M for h in hmms)
    assert Ms[0] <= 60 and any(446 <= m <= 893 for m in Ms) and any(894 <= m <= 1021 for m in Ms) and Ms[-1] > 1500
    assert int(workload[3].max()) > 2000 and int(workload[3].min()) <= 60


def test_batched_cascade_is_bit_exact_and_counts_match_the_oracle(workload, oracle):
    hmms, flat, offsets, lens, _ = workload
    abc = hmms[0].alphabet
    pli = plan7.Pipeline(abc)
    bg = pli.background
    db = plan7.SequenceDatabase.from_packed(abc, flat, offsets, lens)
    oms = [plan7.OptimizedProfile(h, bg, 400) for h in hmms]
    nq, n = len(oms), len(lens)
    xJ = np.zeros((nq, n), dtype=np.int32); xC = np.zeros((nq, n), dtype=np.int32); stage = np.zeros((nq, n), dtype=np.uint8)
    cfg = pli._cfg()
    handles = (C.c_void_p * nq)(*[om._handle for om in oms])
    bgf = np.ascontiguousarray(bg.residue_frequencies, dtype=np.float32)
    st = _lib.lib().p7x_search_batch_raw(C.byref(cfg), handles, nq, bgf.ctypes.data, db._handle, xJ.ctypes.data, xC.ctypes.data, stage.ctypes.data)
    assert st == 0, _lib.last_error()
    # the same batch as a search: per-profile stage counts after the host stage
    hits = list(hmmer.hmmsearch(oms, db, batch=nq))
    pk = easel.PackedBlock.from_arrays(flat, offsets, lens)                  # the oracle's view of the block

    def reference(q):
        op = oracle.OracleProfile(hmms[q], bg, 400)
        recs, ctr = op.cascade_block(pk, want_records=True)
        r = np.frombuffer(recs, dtype=RECORD)
        return (r["xJ_msv"].copy(), r["xC_vit"].copy(), r["ran_vit"].copy(), r["stage"].copy(),
                (ctr.n_past_msv, ctr.n_past_bias, ctr.n_past_vit, ctr.n_past_fwd))

    with ThreadPoolExecutor(max_workers=16) as ex:         # ctypes releases the GIL: one oracle cascade per core
        refs = list(ex.map(reference, range(nq)))
    total_vit = 0
    for q, (xj, xc, ran, stg, counts) in enumerate(refs):
        M = hmms[q].M
        assert np.array_equal(xJ[q], xj), (M, int(np.sum(xJ[q] != xj)))
        on = ran != 0
        assert np.array_equal(xC[q][on], xc[on]), (M, int(np.sum(xC[q][on] != xc[on])))
        assert np.all(xC[q][~on] == np.iinfo(np.int32).min), M
        for k in (1, 2, 3):
            assert int(np.sum(stage[q] >= k)) == counts[k - 1], (M, k)
        assert tuple(hits[q].stage_counts.values()) == counts, M
        survivors = set(np.nonzero(stg == 4)[0].tolist())
        assert {h.seqidx for h in hits[q]} <= survivors, M
        total_vit += int(on.sum())
    assert total_vit > 1000
    assert sum(len(h) for h in hits) > 100, "the planted domains must produce hits"


def _fields(th):
    return [(h.seqidx, h.score, h.pre_score, h.sum_score, h.evalue, h.reported, h.included,
             [(d.env_from, d.env_to, d.score, d.c_evalue, d.i_evalue, d.reported, d.included, d.alignment.target_from, d.alignment.target_to,
               d.alignment.target_sequence) for d in h.domains]) for h in th]


def test_sharded_search_merged_equals_whole(workload):
    hmms, flat, offsets, lens, _ = workload
    abc = hmms[0].alphabet
    n = 20_000                      # per-sequence Python objects: a prefix of the block is enough for the merge semantics
    block = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name=f"t{t:06d}", sequence=flat[offsets[t]:offsets[t] + lens[t]].copy())
                                             for t in range(n)])
    queries = hmms[::3]
    whole = list(hmmer.hmmsearch(queries, block))
    parts = list(hmmer.hmmsearch(queries, block, devices=[0, 0]))
    assert len(whole) == len(parts) == len(queries)
    nhits = 0
    for a, b in zip(parts, whole):
        assert a.Z == b.Z == n and a.domZ == b.domZ
        assert a.stage_counts == b.stage_counts and a.searched_residues == b.searched_residues
        fa, fb = _fields(a), _fields(b)
        # seqidx of a shard's hit is local to the shard: compare by name instead
        assert [h.name for h in a] == [h.name for h in b]
        assert [f[1:] for f in fa] == [f[1:] for f in fb]
        nhits += len(b)
    assert nhits > 20
    # Z-dependent thresholds with model cutoffs left alone (use_bit_cutoffs is per hit; nothing to re-threshold)
    shards = hmmer.make_chunks(block, 2)
    assert abs(sum(len(s) for s in shards[0]) - sum(len(s) for s in shards[1])) < 6000


def test_config3_at_full_size_stage_counts_against_the_oracle(oracle):
    """BASELINE configs[3] at its own size: the bench's 500,000-target block (lognormal lengths up to 5,000,
    every second target with a planted domain of one of the 20,000 library entries) searched by 40 profiles spread evenly
    over the library through hmmer.hmmsearch with its defaults -- the batches, kernel classes and hybrid splits of the `pfam`
    workload.  Per profile the number of targets past the MSV, bias, Viterbi and Forward filters equals the oracle's
    p7_Pipeline restatement over the same 1.75e8 residues (the integer filters are exact, and targets within the F3 guard band are re-decided by the host in the
    reference's summation order), and every hit is a target the oracle lets through Forward."""
    import bench_workloads as bw
    templates = bw.load_templates()
    lengths = bw.library_lengths(LIBRARY)
    cal = bw.Calibrator(templates[0].alphabet)
    entries = list(range(0, LIBRARY, LIBRARY // 40))[:40]
    hmms = [cal.calibrate(bw.make_entry(templates, e, int(lengths[e]))) for e in entries]
    flat, offsets, lens, nplanted = bw.make_targets(500_000, LIBRARY, templates, lengths, planted_frac=0.5)
    abc = hmms[0].alphabet
    bg = plan7.Background(abc)
    db = plan7.SequenceDatabase.from_packed(abc, flat, offsets, lens)
    hits = list(hmmer.hmmsearch(hmms, db))
    pk = easel.PackedBlock.from_arrays(flat, offsets, lens)

    def reference(q):
        op = oracle.OracleProfile(hmms[q], bg, 400)
        recs, ctr = op.cascade_block(pk, want_records=True)
        r = np.frombuffer(recs, dtype=RECORD)
        return (ctr.n_past_msv, ctr.n_past_bias, ctr.n_past_vit, ctr.n_past_fwd), set(np.nonzero(r["stage"] == 4)[0].tolist())

    with ThreadPoolExecutor(max_workers=16) as ex:
        refs = list(ex.map(reference, range(len(hmms))))
    nhits = 0
    for q, (counts, survivors) in enumerate(refs):
        got = tuple(hits[q].stage_counts.values())
        assert got == counts, (entries[q], hmms[q].M, got, counts)
        assert {h.seqidx for h in hits[q]} <= survivors, entries[q]
        nhits += len(hits[q])
    assert nhits > 200
