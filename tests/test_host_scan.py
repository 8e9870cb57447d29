"""The scan orientation's host side without a GPU: per-model results folded into per-sequence hit lists in any order
(p7x_scan_accum_add_indexed) equal the in-order fold (p7x_scan_collect), and hmmer.hmmscan end to end over a stand-in for the
device whose two stages are the oracle's filters and the product's host stage -- against RREFam.scan.tbl."""
import ctypes as C

import numpy as np

import host_pipeline
from conftest import golden_table, load_hmms
from pyhmmer_amd import _lib, hmmer, plan7


def _scan_pipeline(abc, **kw):
    pli = plan7.Pipeline(abc, **kw)
    pli._mode = plan7._P7X_SCAN_MODELS
    return pli


def _per_model(oracle, hmms, block, **kw):
    return [host_pipeline.host_search(oracle, h, block, pipeline=_scan_pipeline(h.alphabet, **kw)) for h in hmms]


def _rows(per_sequence):
    return [[(h.name, round(h.score, 4), h.evalue, h.reported, h.included, [(d.env_from, d.env_to, round(d.score, 4), d.reported) for d in h.domains])
             for h in th] + [th.Z, tuple(th.stage_counts.values()), th.searched_models] for th in per_sequence]


def _accumulate(block, pli, per_model, numbers=None, chunks=1):
    n = len(block)
    names = (C.c_char_p * n)(*[s.name.encode() for s in block])
    lengths = (C.c_int32 * n)(*[len(s) for s in block])
    cfg = pli._cfg()
    acc = C.c_void_p()
    assert _lib.lib().p7x_scan_accum_create(C.byref(cfg), n, names, None, None, lengths, C.byref(acc)) == 0, _lib.last_error()
    order = list(range(len(per_model))) if numbers is None else list(numbers)
    for part in np.array_split(np.arange(len(order)), chunks):
        if len(part) == 0:
            continue
        handles = (C.c_void_p * len(part))(*[per_model[order[int(i)]]._handle for i in part])
        if numbers is None:
            st = _lib.lib().p7x_scan_accum_add(acc, handles, len(part))
        else:
            idx = (C.c_int64 * len(part))(*[order[int(i)] for i in part])
            st = _lib.lib().p7x_scan_accum_add_indexed(acc, handles, idx, len(part))
        assert st == 0, _lib.last_error()
    out = (C.c_void_p * n)()
    assert _lib.lib().p7x_scan_accum_finish(acc, out) == 0, _lib.last_error()
    res = [plan7.TopHits(q, C.c_void_p(out[i])) for i, q in enumerate(block)]
    for r in res:
        r._keep = (names, per_model)
    return res


def test_results_folded_in_any_order_equal_the_fold_in_database_order(libp7x, oracle, proteome):
    """17 models against the proteome at E = 200: the per-model results added with their numbers in a shuffled order, in one to five
    calls, give per-sequence lists identical to the in-order fold -- hits, scores, E-values (Z = number of models), the
    running-Z reportability of every hit (p7_pli_NewModel counts the models seen so far), flags, accounting."""
    hmms = load_hmms("RREFam")[:14] + load_hmms("PF02826") + load_hmms("KR") + load_hmms("Thioesterase")
    block = proteome
    rng = np.random.default_rng(3)
    total = 0
    for E in (200.0, 2e-5):            # the second: weak hits pass at the running Z of an early model and not at the final Z
        per_model = _per_model(oracle, hmms, block, E=E)
        pli = _scan_pipeline(hmms[0].alphabet, E=E)
        in_order = _accumulate(block, pli, per_model)
        want = _rows(in_order)
        # per-sequence accounting: how many models' filters every sequence passed, and the models' nodes
        passed = [sum(tuple(pm.stage_counts.values())[f] for pm in per_model) for f in range(4)]
        assert [sum(tuple(th.stage_counts.values())[f] for th in in_order) for f in range(4)] == passed and passed[0] > 100
        assert all(th.searched_models == len(hmms) and th.searched_nodes == sum(h.M for h in hmms) for th in in_order)
        total += sum(len(r) - 3 for r in want)
        if E < 1.0:
            assert any(not h[3] for r in want for h in r[:-3])
        for chunks in (1, 2, 5):
            numbers = rng.permutation(len(hmms)).tolist()
            assert _rows(_accumulate(block, pli, per_model, numbers=numbers, chunks=chunks)) == want, (E, chunks)
    assert total > 60
    # argument checks
    acc = C.c_void_p()
    n = len(block)
    lengths = (C.c_int32 * n)(*[len(s) for s in block])
    cfg = pli._cfg()
    assert libp7x.p7x_scan_accum_create(C.byref(cfg), n, None, None, None, lengths, C.byref(acc)) == 0
    handles = (C.c_void_p * 1)(per_model[0]._handle)
    assert libp7x.p7x_scan_accum_add_indexed(acc, handles, None, 1) == 11
    assert libp7x.p7x_scan_accum_add_indexed(acc, handles, (C.c_int64 * 1)(-1), 1) == 11
    assert libp7x.p7x_scan_accum_add_indexed(acc, handles, (C.c_int64 * 1)(5), 1) == 0
    assert libp7x.p7x_scan_accum_add_indexed(acc, handles, (C.c_int64 * 1)(5), 1) == 11          # a model number twice
    two = (C.c_void_p * 2)(per_model[0]._handle, per_model[1]._handle)
    assert libp7x.p7x_scan_accum_add_indexed(acc, two, (C.c_int64 * 2)(9, 9), 2) == 11
    libp7x.p7x_scan_accum_destroy(acc)


class _OracleBackedDatabase:
    """Stand-in for hmmer.ShardedDatabase: stage 1 is nothing, stage 2 runs the oracle's filters and parsers and the product's
    host stage for every profile of the batch (tests/host_pipeline.py)."""

    def __init__(self, oracle, block, hmm_of):
        self.oracle, self.block, self.hmm_of = oracle, block, hmm_of
        pk = block.packed()

        class _Shard:
            pass
        sh = _Shard()
        n = len(block)
        sh._names = (C.c_char_p * n)(*[s.name.encode() for s in block])
        sh._accs = (C.c_char_p * n)(*[(s.accession or "").encode() for s in block])
        sh._descs = (C.c_char_p * n)(*[(s.description or "").encode() for s in block])
        sh.block = block
        self.shards = [sh]
        self.total_residues = int(pk.lengths.sum())

    def enqueue(self, pipelines, queries):
        return [(pipelines[0], list(queries))]

    def wait(self, pendings):
        pass

    def abandon(self, pendings):
        pass

    def finish(self, pendings, raw=False):
        pli, queries = pendings[0]
        hits = [host_pipeline.host_search(self.oracle, self.hmm_of[id(q)], self.block, pipeline=pli) for q in queries]
        if not raw:
            return hits
        # the scan orientation asks for the bare handles (plan7.HitHandles): hand them over, the TopHits objects let go of them
        handles = (C.c_void_p * len(hits))(*[h._handle.value for h in hits])
        for h in hits:
            h._handle = None
        return plan7.HitHandles(handles, len(hits))


def test_hmmscan_over_a_stand_in_device_reproduces_the_scan_table(libp7x, oracle, proteome, monkeypatch):
    """hmmer.hmmscan itself -- query blocks, batches in order of model length, the fold of every finished batch with the
    profiles' numbers, the final transposition -- with the device replaced by the oracle + host stage: RREFam.scan.tbl
    (real hmmscan output, reference tests/data/tables) row for row, and the same lists as the in-order fold."""
    hmms = load_hmms("RREFam")
    bg = plan7.Background(hmms[0].alphabet)
    oms = [plan7.OptimizedProfile(h, bg, 400) for h in hmms]
    hmm_of = {id(om): h for om, h in zip(oms, hmms)}
    monkeypatch.setattr(hmmer, "ShardedDatabase", lambda block, devs: _OracleBackedDatabase(oracle, block, hmm_of))
    monkeypatch.setattr(hmmer, "_shard_residues", lambda db: db.total_residues, raising=False)
    monkeypatch.setattr(hmmer, "_batch_cap", lambda db: 7, raising=False)

    class _Lib:                                   # one HIP device "present"; everything else is the real library
        def __getattr__(self, name):
            return (lambda: 1) if name == "p7x_device_count" else getattr(libp7x, name)
    monkeypatch.setattr(hmmer._lib, "lib", lambda: _Lib())
    block = plan7.OptimizedProfileBlock(hmms[0].alphabet, oms)
    got = list(hmmer.hmmscan(proteome, block, batch=7, feeders=2, pipeline_depth=3))
    assert len(got) == len(proteome)
    rows = golden_table("RREFam.scan.tbl")
    by_query = {}
    for r in rows:
        by_query.setdefault(r[2], []).append(r)
    checked = 0
    for seq, hits in zip(proteome, got):
        want = by_query.get(seq.name, [])
        reported = [h for h in hits if h.reported]
        assert [h.name for h in reported] == [r[0] for r in want], seq.name
        for h, r in zip(reported, want):
            assert abs(h.score - float(r[5])) <= 0.051 and abs(h.bias - float(r[6])) <= 0.051, (seq.name, h.name)
            assert abs(h.evalue - float(r[4])) <= 0.06 * float(r[4]), (seq.name, h.name, h.evalue, r[4])
            checked += 1
        assert hits.Z == len(hmms)
    assert checked == len(rows) > 0
    # the same lists as the fold in database order
    monkeypatch.undo()
    per_model = _per_model(oracle, hmms, proteome)
    want = _rows(_accumulate(proteome, _scan_pipeline(hmms[0].alphabet), per_model))
    assert _rows(got) == want


def test_hit_handles_own_their_results_until_they_go(libp7x, oracle, proteome):
    """plan7.HitHandles: the per-model results of a scan batch as the library's bare handles (what
    Pipeline._search_finish_batch(raw=True) returns); p7x_tophits_destroy_many releases them together and zeroes the slots."""
    hmms = load_hmms("RREFam")[:3]
    hits = [host_pipeline.host_search(oracle, h, proteome[:300]) for h in hmms]
    n_before = [len(h) for h in hits]
    handles = (C.c_void_p * len(hits))(*[h._handle.value for h in hits])
    for h in hits:
        h._handle = None                                   # handed over
    batch = plan7.HitHandles(handles, len(hits))
    assert len(batch) == 3 and all(handles[i] for i in range(3))
    assert [int(libp7x.p7x_tophits_nhits(C.c_void_p(handles[i]))) for i in range(3)] == n_before
    del batch
    assert [handles[i] for i in range(3)] == [None, None, None]
    libp7x.p7x_tophits_destroy_many(handles, 3)            # empty slots are fine
    libp7x.p7x_tophits_destroy_many(None, 0)
