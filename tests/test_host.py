"""Host logic of the product (no GPU): file parsing, profile conversion, packing, ABI surface, error behaviour."""
import ctypes as C
import re
import sys

import numpy as np
import pytest

import h3_reader
from conftest import GOLDEN, ROOT, load_hmms, synthetic_block
from pyhmmer_amd import _lib, easel, errors, hmmer, plan7


def test_abi_exports_every_declared_symbol(libp7x):
    header = (ROOT / "include" / "p7x.h").read_text()
    declared = set(re.findall(r"\b(p7x_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes found in include/p7x.h"
    raw = C.CDLL(str(_lib.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(raw, name), f"libp7x.so does not export {name}"
    assert declared == set(_lib.declared_symbols()), declared ^ set(_lib.declared_symbols())
    assert libp7x.p7x_abi_version() == 8


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam"])
def test_product_conversion_bit_exact_vs_pressed(name, models):
    f = h3_reader.read_h3f(GOLDEN / "db" / f"{name}.hmm.h3f")
    p = h3_reader.read_h3p(GOLDEN / "db" / f"{name}.hmm.h3p")
    for hmm, ff, pp in zip(models[name], f, p):
        om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 400)
        assert np.array_equal(om.rbv, ff["rbv"]) and np.array_equal(om.sbv, ff["sbv"])
        assert np.array_equal(om.rwv, pp["rwv"]) and np.array_equal(om.twv, pp["twv"])
        assert np.array_equal(om.rfv.view(np.uint32), pp["rfv"].view(np.uint32))
        assert np.array_equal(om.tfv.view(np.uint32), pp["tfv"].view(np.uint32))
        assert (om.tbm, om.tec, om.tjb, om.base, om.bias) == (ff["tbm"], ff["tec"], ff["tjb"], ff["base"], ff["bias"])
        assert np.array_equal(om.xw, pp["xw"]) and om.ddbound_w == pp["ddbound_w"]
        assert np.array_equal(om.xf.view(np.uint32), pp["xf"].view(np.uint32))
        assert np.allclose(om.compositions, ff["compo"][: hmm.alphabet.K])
        assert np.allclose(om.evalue_parameters.as_vector(), ff["evparam"])
        assert om.M == ff["M"] and hmm.name == ff["name"] == pp["name"]


def test_hmm_parser_fields(models):
    hmm = models["PF02826"][0]
    assert (hmm.name, hmm.accession, hmm.M) == ("2-Hacid_dh_C", "PF02826.20", 178)
    assert hmm.cutoffs.gathering == (pytest.approx(25.1), pytest.approx(25.1))
    assert hmm.evalue_parameters.m_mu == pytest.approx(-10.2529)
    assert hmm.composition is None                       # no COMPO line
    assert len(hmm.consensus) == hmm.M
    assert np.allclose(hmm.match_emissions[1:].sum(axis=1), 1.0, atol=1e-4)
    assert np.allclose(hmm.transition_probabilities[1:-1, :3].sum(axis=1), 1.0, atol=1e-4)
    rre = models["RREFam"][0]
    assert rre.composition is not None and abs(float(rre.composition.sum()) - 1.0) < 1e-3
    assert len(models["RREFam"]) == 10


def test_alphabet_and_packing(proteome):
    abc = easel.Alphabet.amino()
    assert (abc.K, abc.Kp, abc.symbols) == (20, 29, "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~")
    assert abc.decode(abc.encode("ACDXZ*")) == "ACDXZ*"
    with pytest.raises(ValueError):
        abc.encode("AC1")
    assert len(proteome) == 2100 and proteome.total_length() == 682583      # fixture facts, SURVEY.md Appendix A
    pk = proteome.packed()
    assert pk.dsq[0] == 255 and pk.dsq[-1] == 255
    for t in (0, 1, 17, 2099):
        s = proteome[t]
        o, L = int(pk.offsets[t]), int(pk.lengths[t])
        assert L == len(s) and np.array_equal(pk.dsq[o:o + L], s.sequence)
        assert pk.dsq[o - 1] == 255 and pk.dsq[o + L] == 255


def test_make_chunks_balanced_by_residues(proteome):
    chunks = hmmer.make_chunks(proteome, 8)
    assert sum(len(c) for c in chunks) == len(proteome)
    names = [s.name for c in chunks for s in c]
    assert names == [s.name for s in proteome]              # contiguous, order preserving
    res = [c.total_length() for c in chunks]
    assert max(res) - min(res) < 2 * max(len(s) for s in proteome)


def test_pipeline_argument_validation():
    abc = easel.Alphabet.amino()
    with pytest.raises(errors.InvalidParameter):
        plan7.Pipeline(abc, bit_cutoffs="nope")
    with pytest.raises(errors.InvalidParameter):
        plan7.Pipeline(abc, F1=1.5)
    with pytest.raises(errors.AlphabetMismatch):
        plan7.Pipeline(abc, background=plan7.Background(easel.Alphabet.dna()))
    p = plan7.Pipeline(abc, Z=100, T=5.0)
    c = p._cfg()
    assert (c.Z, c.Z_setby, c.by_E, c.T) == (100.0, 1, 0, 5.0)
    d = plan7.Pipeline(abc)._cfg()
    assert (d.F1, d.F2, d.F3, d.E, d.incE, d.seed) == (0.02, 1e-3, 1e-5, 10.0, 0.01, 42)   # plan7.pyx:5413-5421


def test_no_cpu_fallback_without_a_device(libp7x, models):
    """On a box without a GPU every device entry point must fail loudly (never compute on the CPU)."""
    if libp7x.p7x_device_count() > 0:
        pytest.skip("a HIP device is present")
    hmm = models["PF02826"][0]
    om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 100)
    blk = synthetic_block(4, 50, 1)
    with pytest.raises(errors.DeviceUnavailable):
        om.msv_filter(blk[0])
    with pytest.raises(errors.DeviceUnavailable):
        plan7.Pipeline(hmm.alphabet).search_hmm(hmm, blk)
    with pytest.raises(errors.DeviceUnavailable):
        list(hmmer.hmmsearch(hmm, blk))
    with pytest.raises(errors.DeviceUnavailable):
        list(hmmer.hmmscan(blk, [hmm]))
    with pytest.raises(errors.DeviceUnavailable):
        plan7.Pipeline(hmm.alphabet).scan_seq(blk[0], [hmm])


def test_scan_collect_and_pending_argument_checks(libp7x):
    """Host-only entry points of the scan / two-stage API reject bad arguments instead of crashing."""
    cfg = _lib.PipelineCfg()
    libp7x.p7x_pipeline_cfg_default(C.byref(cfg))
    out = (C.c_void_p * 2)()
    assert libp7x.p7x_scan_collect(None, 0, C.byref(cfg), 2, None, None, None, None, out) == 11
    lengths = (C.c_int32 * 2)(5, 7)
    names = (C.c_char_p * 2)(b"a", b"b")
    handles = (C.c_void_p * 1)()
    assert libp7x.p7x_scan_collect(handles, 0, C.byref(cfg), 2, names, None, None, lengths, out) == 0     # no models: empty lists
    for i in range(2):
        assert libp7x.p7x_tophits_nhits(out[i]) == 0
        libp7x.p7x_tophits_destroy(out[i])
    assert libp7x.p7x_scan_collect(handles, 1, C.byref(cfg), 2, names, None, None, lengths, out) == 11    # NULL per-model result
    libp7x.p7x_pending_destroy(None)
    res = C.c_void_p()
    assert libp7x.p7x_search_block_finish(None, None, None, None, C.byref(res)) == 11


def test_sequence_length_limit(libp7x):
    abc = easel.Alphabet.amino()
    big = easel.DigitalSequence(abc, name="big", sequence=np.zeros(100001, dtype=np.uint8))
    with pytest.raises(ValueError):
        plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, [big]))


def test_msv_isa_never_touches_a_vgpr_with_an_lds_load_in_flight(tmp_path):
    """The MSV kernels issue their LDS loads from inline asm and count completions by hand, which the compiler
    cannot check.  Compile the kernels to ISA and verify statically that no instruction reads or writes a register
    whose ds_read is still outstanding (a missing early-clobber once let a load's result overwrite the address of
    the next load: a timing-dependent wrong score in 4e-4 of targets)."""
    import shutil
    import subprocess
    import sys
    from pathlib import Path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    root = Path(__file__).resolve().parents[1]
    from pyhmmer_amd import _lib
    asm = _lib.fresh_isa("p7x_msv.hip")                 # what build() compiled, if it is current; else compile the unit here
    if asm is None:
        asm = tmp_path / "msv.s"
        subprocess.run([hipcc, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-S",
                        "--cuda-device-only", "-I", str(root / "include"), "-o", str(asm), str(root / "pyhmmer_amd/csrc/p7x_msv.hip")],
                       check=True, capture_output=True)
    r = subprocess.run([sys.executable, str(root / "scripts/check_lds_asm.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "ds_read_b64" in asm.read_text()


def test_envelope_kernel_isa_keeps_memory_round_trips_out_of_its_row_loops(tmp_path):
    """Two properties of the envelope kernel that only show in the ISA and were worth 18 % of its time:
    its phase fences are work-group scope (agent scope is `buffer_wbl2` / `buffer_inv sc1` on gfx950: the whole L2 written
    back and invalidated per phase and wavefront), and no row loop waits for a load behind a store of the same iteration
    (`vmcnt` retires in order): Forward's and Backward's loops have no load at all, the decoding loop waits once, ahead of
    its stores.  scripts/isa_inner_loops.py reads the loop bodies from the compiler's block annotations."""
    import shutil
    import subprocess
    import sys
    from pathlib import Path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    root = Path(__file__).resolve().parents[1]
    from pyhmmer_amd import _lib
    asm = _lib.fresh_isa("p7x_envelope.hip")
    if asm is None:
        asm = tmp_path / "env.s"
        subprocess.run([hipcc, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-S",
                        "--cuda-device-only", "-I", str(root / "include"), "-o", str(asm), str(root / "pyhmmer_amd/csrc/p7x_envelope.hip")],
                       check=True, capture_output=True)
    text = asm.read_text()
    assert "buffer_wbl2" not in text and "buffer_inv" not in text
    sys.path.insert(0, str(root / "scripts"))
    try:
        import isa_inner_loops
    finally:
        sys.path.pop(0)
    for inst in ("ILi5ELb1ELb0E", "ILi3ELb1ELb0E", "ILi12ELb1ELb0E"):          # the benchmark's class, a short and a long one
        loops = isa_inner_loops.loops(asm, "env_kernel" + inst)
        rows = [l for l in loops if l[4] >= 2 and l[2] > 200]           # row loops (the traceback walk, a pointer chase, is shorter)
        assert len(rows) >= 3, loops
        assert not any(serial for *_, serial in rows), rows
        assert sum(1 for l in rows if l[3] == 0) >= 2, rows            # Forward and Backward: stores only


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam"])
def test_pressed_files_are_written_byte_for_byte(name, models, tmp_path):
    """hmmpress: .h3f / .h3p produced here == the files real HMMER pressed (reference tests/data/hmms/db), byte for
    byte; hmmpress as a whole (which also writes the .h3m the offsets point into) reproduces them too."""
    want_f = (GOLDEN / "db" / f"{name}.hmm.h3f").read_bytes()
    want_p = (GOLDEN / "db" / f"{name}.hmm.h3p").read_bytes()
    offs = [r["offs"] for r in h3_reader.read_h3f(GOLDEN / "db" / f"{name}.hmm.h3f")]
    import io
    ff, fp = io.BytesIO(), io.BytesIO()
    for hmm, o in zip(models[name], offs):
        om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 400)
        om.write(ff, fp, offsets=(int(o[0]), ff.tell(), fp.tell()))
        assert (ff.tell(), fp.tell()) != (0, 0)
    assert ff.getvalue() == want_f
    assert fp.getvalue() == want_p
    assert hmmer.hmmpress(models[name], tmp_path / "db") == len(models[name])
    assert (tmp_path / "db.h3f").read_bytes() == want_f and (tmp_path / "db.h3p").read_bytes() == want_p
    assert (tmp_path / "db.h3m").read_bytes() == (GOLDEN / "hmms" / f"{name}.h3m").read_bytes()


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam"])
def test_pressed_files_are_read_back_exactly(name, models):
    """HMMPressedFile: every table of every record equals the profile converted from the text model."""
    with plan7.HMMPressedFile(GOLDEN / "db" / f"{name}.hmm") as pressed:
        assert len(pressed) == len(models[name])
        for om, hmm in zip(pressed, models[name]):
            ref = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 400)
            assert (om.name, om.accession, om.description) == (hmm.name, hmm.accession, hmm.description)
            assert om.consensus == hmm.consensus
            assert (om.M, om.L, om.tbm, om.tec, om.tjb, om.base, om.bias) == (ref.M, 400, ref.tbm, ref.tec, ref.tjb, ref.base, ref.bias)
            for tab in ("rbv", "sbv", "rwv", "twv", "rfv", "tfv"):
                a, b = getattr(om, tab), getattr(ref, tab)
                assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), tab
            assert np.array_equal(om.evalue_parameters.as_vector(), ref.evalue_parameters.as_vector()) if hasattr(om, "evalue_parameters") else True
    with pytest.raises(FileNotFoundError):
        plan7.HMMPressedFile(GOLDEN / "db" / "nothing.hmm")


def test_native_fasta_reader_equals_python_reader(tmp_path):
    """SequenceFile.read_block() goes through p7x_fasta_parse for digital files; record by record it must give what
    the per-record Python reader gives (names, descriptions, residues), and reject illegal symbols."""
    path = GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa"
    abc = easel.Alphabet.amino()
    with easel.SequenceFile(path, digital=True, alphabet=abc) as f:
        fast = f.read_block()
        assert f.read() is None                                   # consumed
    with easel.SequenceFile(path, digital=True, alphabet=abc) as f:
        slow = easel.DigitalSequenceBlock(abc, list(iter(f.read, None)))
    assert type(fast).__name__ == "_LazyDigitalSequenceBlock" and len(fast) == len(slow) == 2100
    assert fast.total_length() == slow.total_length()
    pk_fast, pk_slow = fast.packed(), slow.packed()
    assert np.array_equal(pk_fast.dsq, pk_slow.dsq) and np.array_equal(pk_fast.offsets, pk_slow.offsets)
    assert np.array_equal(pk_fast.lengths, pk_slow.lengths)
    for t in (0, 1, 17, 2099, -1):
        a, b = fast[t], slow[t]
        assert (a.name, a.description) == (b.name, b.description) and np.array_equal(a.sequence, b.sequence)
    assert [s.name for s in fast[10:13]] == [s.name for s in slow[10:13]]      # slicing materialises the list
    odd = tmp_path / "odd.fa"
    odd.write_text("\n>s1  two  words \r\nAC DE\nfg 10\n>s2\n\n>s3 x\nWWW")
    with easel.SequenceFile(odd, digital=True, alphabet=abc) as f:
        blk = f.read_block()
    assert [(s.name, s.description, len(s)) for s in blk] == [("s1", "two  words", 6), ("s2", "", 0), ("s3", "x", 3)]
    bad = tmp_path / "bad.fa"
    bad.write_text(">s1\nAC?DE\n")
    with easel.SequenceFile(bad, digital=True, alphabet=abc) as f:
        with pytest.raises(ValueError):
            f.read_block()


class _FakeShards:
    """Stands in for hmmer.ShardedDatabase in the query pipeline: stage 1 of a batch of queries 'runs' for a random time
    after enqueue.  A batch fails when it contains a failing query."""

    def __init__(self, seed, fail_at=(), fail_in="enqueue"):
        import random
        self.rnd = random.Random(seed)
        self.fail_at, self.fail_in = set(fail_at), fail_in
        self.abandoned = []

    def _check(self, qs, where):
        bad = [q for q in qs if q in self.fail_at]
        if bad and self.fail_in == where:
            raise ValueError(f"query {bad[0]}")

    def enqueue(self, pipelines, qs):
        self._check(qs, "enqueue")
        import time
        return [("pend", list(qs), time.perf_counter() + self.rnd.random() * 0.004)]

    def wait(self, pendings):
        import time
        _, qs, due = pendings[0]
        while time.perf_counter() < due:
            time.sleep(0.0002)
        self._check(qs, "wait")

    def abandon(self, pendings):
        self.abandoned.extend(pendings or ())

    def finish(self, pendings):
        import time
        time.sleep(self.rnd.random() * 0.002)
        self._check(pendings[0][1], "finish")
        return [("hits", q) for q in pendings[0][1]]

    def search(self, pipelines, qs):
        p = self.enqueue(pipelines, qs)
        self.wait(p)
        return self.finish(p)


@pytest.mark.timeout(120)
def test_query_pipeline_orders_results_and_errors_for_every_shape(libp7x, monkeypatch):
    """hmmer._run_queries (the overlap of consecutive queries behind hmmsearch / hmmscan) with a stand-in for the
    device: results in query order, an error surfaces at its query's position after the results before it, an
    abandoned generator leaves no thread behind -- for many feeder / depth / window / finisher combinations."""
    import threading
    from pyhmmer_amd import hmmer
    before = threading.active_count()
    shapes = [(0, 1, 1, 0), (1, 1, 1, 0), (2, 1, 1, 0), (4, 2, 1, 0), (8, 4, 1, 0), (8, 1, 8, 0), (32, 2, 8, 0), (5, 3, 2, 0),
              (2, 2, 8, 4), (64, 4, 4, 8), (3, 8, 64, 2), (4, 2, 1, 4),      # ... the shape bench.py ran the headline with until round 5
              (8, 0, 1, 0)]                                                   # feeders = 0: the first batch decides (hmmsearch's default)
    for seed, (depth, feeders, window, fin) in enumerate(shapes):
        batch = (1, 3, 8)[seed % 3]             # queries per device batch
        for n in (0, 1, 7, 40):
            db = _FakeShards(seed)
            out = list(hmmer._run_queries(db, [None], iter(range(n)), depth, feeders, window, fin, batch=batch))
            assert out == [(q, ("hits", q)) for q in range(n)], (depth, feeders, window, fin, n)
        for where in ("enqueue", "wait", "finish"):
            db = _FakeShards(seed, fail_at={11}, fail_in=where)
            got = []
            with pytest.raises(ValueError, match="query 11"):
                for q, h in hmmer._run_queries(db, [None], iter(range(30)), depth, feeders, window, fin, batch=batch):
                    got.append(q)
            # the error surfaces at the failing query's position, after the results of every query before it (a batch
            # with a failing member is repeated one query at a time)
            assert got == list(range(11)), (depth, feeders, window, fin, where, got)
        gen = hmmer._run_queries(_FakeShards(seed), [None], iter(range(1000)), depth, feeders, window, fin)
        assert [next(gen)[0] for _ in range(3)] == [0, 1, 2]
        gen.close()                                        # abandoned: feeders stop, queued work is released

        def failing():
            yield 0
            yield 1
            raise RuntimeError("iterable failed")
        with pytest.raises(RuntimeError, match="iterable failed"):
            list(hmmer._run_queries(_FakeShards(seed), [None], failing(), depth, feeders, window, fin))
    deadline = __import__("time").time() + 5
    while threading.active_count() > before and __import__("time").time() < deadline:
        __import__("time").sleep(0.05)
    assert threading.active_count() <= before


def test_the_first_batch_decides_the_feeders(libp7x):
    """hmmsearch(feeders=0), the default: batches of several different profiles run with three cascades in flight, batches of
    one profile (a single query, a stream of the same one) with two (hmmer._run_batches; measured: profiles/r06_feeders.txt)."""
    from pyhmmer_amd import hmmer

    class Q:
        def __init__(self, M):
            self.M = M

    def feeders_of(queries, batch):
        out = list(hmmer._run_queries(_FakeShards(1), [None], iter(queries), 8, 0, 1, 0, batch=batch))
        assert len(out) == len(queries)
        return hmmer.pipeline_stats()["feeders"]

    same = Q(262)
    assert feeders_of([same] * 40, 7) == 2                       # the one-profile stream of configs[1]
    assert feeders_of([Q(100 + i) for i in range(40)], 8) == 3   # a profile library
    assert feeders_of([Q(300)], 4) == 2                          # a single query
    assert feeders_of([], 4) == 2


def test_automatic_batches_follow_the_cell_budget():
    """batch=0: queries are sorted by model length inside a span and cut so that a batch holds at most the cell budget
    (short models: many per batch, up to 256; long ones: few), results still come back in input order."""
    import random
    from pyhmmer_amd import hmmer

    class Q:
        def __init__(self, i, M):
            self.i, self.M = i, M

    class Recorder(_FakeShards):
        shard_residues = 100_000_000
        shard_targets = 300_000

        def __init__(self):
            super().__init__(3)
            self.batches = []

        def enqueue(self, pipelines, qs):
            self.batches.append([q.M for q in qs])
            return super().enqueue(pipelines, qs)

    rnd = random.Random(5)
    qs = [Q(i, rnd.choice((40, 90, 150, 400, 1200, 3000))) for i in range(700)]
    db = Recorder()
    out = list(hmmer._run_queries(db, [None], iter(qs), 3, 2, 1, 0, batch=0))
    assert [q.i for q, _ in out] == list(range(700))
    assert sum(len(b) for b in db.batches) == 700
    budget = hmmer._BATCH_CELLS
    for b in db.batches:
        assert 1 <= len(b) <= hmmer._BATCH_MAX and b == sorted(b)
        assert len(b) == 1 or sum(b) * Recorder.shard_residues <= budget * (1 + 1e-9)
    assert 64 < max(len(b) for b in db.batches) <= int(budget // (40 * Recorder.shard_residues))    # 40-node models: the budget holds 150 of them
    assert any(b == [3000, 3000] for b in db.batches)                    # 3e11 cells each: two per batch
    # a small block (hmmscan's query sequences, a proteome): the cap grows to what the workspace allows
    small = Recorder()
    small.shard_residues, small.shard_targets = 700_000, 2_100
    assert hmmer._batch_cap(small) == 3994 and hmmer._batch_cap(db) == hmmer._BATCH_MAX
    out = list(hmmer._run_queries(small, [None], qs, 3, 2, 1, 0, batch=0))          # a sized source: sorted as a whole
    assert [q.i for q, _ in out] == list(range(700)) and len(small.batches) == 1 and small.batches[0] == sorted(q.M for q in qs)


def test_forward_parser_in_reference_summation_order_is_bit_identical_to_the_oracle(models, oracle, proteome):
    """The F3 tie-breaker (p7x_forward_parser_exact) must be the reference's Forward parser bit for bit: it is
    compared with the SSE2 restatement of impl_sse/fwdback.c in oracle/ on short (M < 100: three fixed carry sweeps) and
    long (conditional sweeps) models, including targets long enough to rescale."""
    import ctypes as C
    import numpy as np
    from pyhmmer_amd import _lib, plan7
    lib = _lib.lib()
    rng = np.random.default_rng(7)
    for name, idx in (("RREFam", 0), ("PF02826", 0), ("LuxC", 0)):
        hmm = models[name][idx]
        bg = plan7.Background(hmm.alphabet)
        op = oracle.OracleProfile(hmm, bg, 400)
        om = plan7.OptimizedProfile(hmm, bg, 400)
        seqs = [proteome[i] for i in rng.choice(len(proteome), size=25, replace=False)]
        seqs.sort(key=len)
        for s in seqs[:12] + seqs[-3:]:
            st, want = op.fwd(s.sequence)
            d = np.concatenate([[255], np.asarray(s.sequence, dtype=np.uint8), [255]]).astype(np.uint8)
            got = C.c_float()
            st2 = lib.p7x_forward_parser_exact(om._handle, d.ctypes.data, len(s), C.byref(got))
            assert (st == 0) == (st2 == 0)
            if st == 0:
                assert np.float32(got.value).tobytes() == np.float32(want).tobytes(), (name, s.name, got.value, want)


def test_f3_guard_follows_the_reference_order_on_the_threshold(models, oracle, proteome):
    """A target whose Forward P-value is put exactly on F3 (and one ulp beside it) is kept / dropped as the reference's
    arithmetic says, whatever score the first stage handed over: the first stage lets the guard band through, the host
    stage re-scores the targets inside it in the reference's summation order."""
    import math
    import numpy as np
    import host_pipeline
    from pyhmmer_amd import plan7
    hmm = models["PF02826"][0]
    bg = plan7.Background(hmm.alphabet)
    block = proteome[:700]
    op = oracle.OracleProfile(hmm, bg, 400)
    recs, _ = op.cascade_block(block.packed(), F3=1.0)             # every Viterbi survivor gets a Forward score
    ftau, flam = float(hmm.evalue_parameters.f_tau), float(hmm.evalue_parameters.f_lambda)

    def pval(rec):
        sc = np.float32((np.float64(np.float32(rec.fwdsc) - np.float32(rec.filtersc))) / 0.69314718055994529)
        return math.exp(-flam * (float(sc) - ftau)) if float(sc) >= ftau else 1.0

    cand = sorted((t for t in range(len(block)) if recs[t].stage >= 4 and 1e-9 < pval(recs[t]) < 1e-2), key=lambda t: pval(recs[t]))
    assert cand
    t = cand[len(cand) // 2]
    P = pval(recs[t])
    for F3, kept in ((P, True), (float(np.nextafter(P, 0.0)), False)):
        want = sum(1 for u in range(len(block)) if recs[u].stage >= 4 and not (pval(recs[u]) > F3))
        for ulps in (-3, 3):                                       # the first stage's score is a few ulps off either way
            off = np.float32(recs[t].fwdsc)
            for _ in range(abs(ulps)):
                off = np.nextafter(off, np.float32(1e30 if ulps > 0 else -1e30))
            pli = plan7.Pipeline(hmm.alphabet, F3=F3)
            hits = host_pipeline.host_search(oracle, hmm, block, pipeline=pli, F=(0.02, 1e-3, F3 * (1.0 + 4e-3)), perturb_fwd={t: float(off)})
            assert hits.stage_counts["fwd"] == want, (F3, kept, ulps)


def test_integer_thresholds_of_a_traceback_choice_are_the_reference_test(libp7x):
    """p7x_choice.hpp restates esl_rnd_FChoose's floating-point test (roll < cumulative / norm, roll = x / 2^32) as integer
    thresholds on the generator's state: the same path for every weight vector and every state, also next to a threshold."""
    import ctypes as C
    rng = np.random.default_rng(5)
    a, b = C.c_int(0), C.c_int(0)
    nchecked = 0
    for trial in range(3000):
        n = int(rng.choice([2, 4]))
        kind = trial % 6
        if kind == 0:
            p = rng.random(n)
        elif kind == 1:
            p = rng.random(n) * 10.0 ** rng.integers(-30, 5, size=n)
        elif kind == 2:
            p = rng.random(n); p[rng.integers(0, n)] = 0.0
        elif kind == 3:
            p = np.zeros(n); p[rng.integers(0, n)] = rng.random()
        elif kind == 4:
            p = np.zeros(n)                                   # esl_vec_FNorm's uniform fallback
        else:
            p = rng.random(n); p[-1] = 0.0; p[0] = 1e-30
        p32 = np.ascontiguousarray(p, dtype=np.float32)
        s = float(p32.sum()) or 1.0
        cum = np.cumsum(p32.astype(np.float64)) / s
        xs = [0, 1, 2**32 - 1, 2**32 - 2, 2**31] + [int(v) for v in rng.integers(0, 2**32, size=6)]
        for c in cum:                                         # states around every threshold
            t = int(min(max(np.ceil(c * 2.0**32), 0), 2**32 - 1))
            xs += [min(max(t + d, 0), 2**32 - 1) for d in (-3, -2, -1, 0, 1, 2, 3, 300, -300)]
        for x in xs:
            assert libp7x.p7x_debug_choice(p32.ctypes.data, n, C.c_uint32(x), C.byref(a), C.byref(b)) == 0
            assert a.value == b.value, (p32, x, a.value, b.value)
            nchecked += 1
    assert nchecked > 50000


@pytest.mark.parametrize("name", ["Thioesterase", "PF02826", "RREFam", "RF00001"])
def test_binary_hmm_files_read_like_their_text_twins(libp7x, name, tmp_path):
    """HMMFile reads the binary 3/f format (.h3m; reference plan7.pyx:3656-3760, HMMFile's own docstring example:
    tests/data/hmms/bin/RREFam.h3m holds 10 HMMs, the first one RREFam002.1).  The reference's four binary fixtures against
    the text files of the same models: every field equal, the probabilities bit for bit (the binary format stores the float
    the text parser computes)."""
    import io
    with plan7.HMMFile(GOLDEN / "hmms" / f"{name}.h3m") as f:
        binary = list(f)
    with plan7.HMMFile(GOLDEN / "hmms" / f"{name}.hmm") as f:
        text = list(f)
    assert len(binary) == len(text) == (10 if name == "RREFam" else 1)
    if name == "RREFam":
        assert binary[0].accession == "RREFam002.1"
    for b, t in zip(binary, text):
        for attr in ("M", "name", "accession", "description", "consensus", "reference", "model_mask", "consensus_structure", "max_length",
                     "nseq", "checksum", "command_line", "creation_time"):
            assert getattr(b, attr) == getattr(t, attr), (name, attr)
        assert b.alphabet == t.alphabet
        assert np.array_equal(b.transition_probabilities, t.transition_probabilities)
        assert np.array_equal(b.match_emissions, t.match_emissions) and np.array_equal(b.insert_emissions, t.insert_emissions)
        assert np.array_equal(b._evparam, t._evparam) and np.array_equal(b._cutoffs, t._cutoffs)
        assert (b.composition is None) == (t.composition is None) and (b.composition is None or np.array_equal(b.composition, t.composition))
        assert (b.map is None) == (t.map is None) and (b.map is None or np.array_equal(b.map, t.map))
        assert b.nseq_effective == t.nseq_effective          # both hold upstream's float: text and binary forms compare equal
        assert b == t
    # a text-format file behind a BINARY handle (what the reference's HMMFile takes), an older magic behind a handle
    text_path = GOLDEN / "hmms" / f"{name}.hmm"
    with open(text_path, "rb") as fh:
        assert [h.name for h in plan7.HMMFile(fh)] == [h.name for h in text]
    assert [h.name for h in plan7.HMMFile(io.BytesIO(text_path.read_bytes()))] == [h.name for h in text]
    with pytest.raises(ValueError, match="older format"):
        plan7.HMMFile(io.BytesIO((0xe8ededb8).to_bytes(4, sys.byteorder) + b"\0" * 64))
    # a file object, rewinding, a truncated file, an older binary version
    raw = (GOLDEN / "hmms" / f"{name}.h3m").read_bytes()
    with plan7.HMMFile(io.BytesIO(raw)) as f:
        again = list(f)
        f.rewind()
        assert [h.name for h in f] == [h.name for h in binary] == [h.name for h in again]
    (tmp_path / "cut.h3m").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(ValueError, match="premature end"):
        list(plan7.HMMFile(tmp_path / "cut.h3m"))
    (tmp_path / "old.h3m").write_bytes((0xe8ededb8).to_bytes(4, sys.byteorder) + raw[4:])
    with pytest.raises(ValueError, match="older format"):
        plan7.HMMFile(tmp_path / "old.h3m")


@pytest.mark.parametrize("name", ["PF02826", "RF00001", "RREFam", "Thioesterase", "KR", "LuxC", "bmyD"])
def test_hmm_write_reproduces_the_save_files(libp7x, name):
    """HMM.write (reference plan7.pyx:3403-3436): the text form of a parsed model is the fixture it was parsed from, line for
    line behind the version banner, and reading the binary form back gives the same model; the binary form of the four models
    that have a .h3m fixture is that fixture, byte for byte."""
    import io
    path = GOLDEN / "hmms" / f"{name}.hmm"
    with plan7.HMMFile(path) as f:
        hmms = list(f)
    out = io.StringIO()
    for h in hmms:
        h.write(out)
    got, want = out.getvalue().splitlines(), path.read_text().splitlines()      # (one fixture ends without a newline)
    assert len(got) == len(want) and out.getvalue().endswith("//\n")
    assert [g for g, w in zip(got, want) if g != w and not w.startswith("HMMER3/f")] == []
    raw = io.BytesIO()
    for h in hmms:
        h.write(raw, binary=True)
    if (GOLDEN / "hmms" / f"{name}.h3m").exists():
        assert raw.getvalue() == (GOLDEN / "hmms" / f"{name}.h3m").read_bytes()
    raw.seek(0)
    back = list(plan7.HMMFile(raw))
    assert [b.name for b in back] == [h.name for h in hmms]
    for b, h in zip(back, hmms):
        assert np.array_equal(b.match_emissions, h.match_emissions) and np.array_equal(b.transition_probabilities, h.transition_probabilities)
        assert np.array_equal(b._evparam, h._evparam) and b.consensus == h.consensus and b.checksum == h.checksum
    tb = io.BytesIO()
    hmms[0].write(tb)                                  # a binary handle takes the text form as bytes
    assert tb.getvalue().decode() == hmms[0]._to_text()


def test_hmmfile_finds_the_pressed_database_beside_a_path(libp7x, models, tmp_path):
    """HMMFile(path, db=True) (reference plan7.pyx:3684-3760, 4009-4048): a pressed database beside the path is used -- models from
    <path>.h3m, optimized_profiles() over <path>.h3f / .h3p -- and db=False ignores it; `closed`, repr, and the errors of opening
    what is not an HMM file.  (The reference's doctest: bin/Thioesterase.h3m is not pressed, db/Thioesterase.hmm is.)"""
    assert plan7.HMMFile(GOLDEN / "hmms" / "Thioesterase.h3m").is_pressed() is False
    base = tmp_path / "RREFam.hmm"
    assert hmmer.hmmpress(models["RREFam"], base) == 10
    (tmp_path / "RREFam.hmm").write_text((GOLDEN / "hmms" / "RREFam.hmm").read_text())
    with plan7.HMMFile(base) as f:
        assert f.is_pressed() and not f.closed and repr(f) == f"HMMFile({str(base)!r})"
        from_h3m = list(f)                                  # read from the binary file of the database
        oms = list(f.optimized_profiles())
    assert f.closed
    with pytest.raises(ValueError, match="closed file"):
        f.read()
    with pytest.raises(ValueError, match="closed file"):
        f.is_pressed()
    assert [h.name for h in from_h3m] == [h.name for h in models["RREFam"]] == [om.name for om in oms]
    assert all(np.array_equal(a.match_emissions, b.match_emissions) for a, b in zip(from_h3m, models["RREFam"]))
    with plan7.HMMPressedFile(base) as direct:
        assert [om.name for om in direct] == [om.name for om in oms]
    with plan7.HMMFile(base, db=False) as f:                # the text file itself
        assert not f.is_pressed() and len(list(f)) == 10
        with pytest.raises(ValueError, match="does not contain optimized profiles"):
            f.optimized_profiles()
    # only the pressed files, no text file of that name: still a database (upstream opens <name>.h3m first)
    (tmp_path / "RREFam.hmm").unlink()
    with plan7.HMMFile(base) as f:
        assert f.is_pressed() and len(list(f)) == 10
    with pytest.raises(FileNotFoundError):
        plan7.HMMFile(tmp_path / "nothing.hmm")
    with pytest.raises(IsADirectoryError):
        plan7.HMMFile(tmp_path)
    (tmp_path / "empty.hmm").write_bytes(b"")
    with pytest.raises(EOFError):
        plan7.HMMFile(tmp_path / "empty.hmm")
    (tmp_path / "junk.hmm").write_text(">seq1\nACGT\n")
    with pytest.raises(ValueError, match="not recognized"):
        plan7.HMMFile(tmp_path / "junk.hmm")


def test_hmm_statistics_and_edits(libp7x, models):
    """The HMM's own methods beside the save formats (reference plan7.pyx:2418-2560, 3124-3140, 3335-3655 and
    tests/test_plan7/test_hmm.py:48-252).  Known answers: the three doctest values of the reference for the Thioesterase model
    (`mean_match_entropy` 3.0425, `mean_match_information` 1.1330, `mean_match_relative_entropy` 1.1201); `set_composition` and
    `set_consensus` against the COMPO and consensus lines that real hmmbuild wrote into the fixtures."""
    import copy
    import pickle
    th = models["Thioesterase"][0]
    bg = plan7.Background(th.alphabet)
    assert th.mean_match_entropy() == pytest.approx(3.0425, abs=5e-5)
    assert th.mean_match_information(bg) == pytest.approx(1.1330, abs=5e-5)
    assert th.mean_match_relative_entropy(bg) == pytest.approx(1.1201, abs=1e-4)
    for name in ("LuxC", "RREFam"):
        for hmm in models[name]:
            again = hmm.copy()
            again.composition, again.consensus = None, None
            again.set_composition()
            again.set_consensus()
            assert np.allclose(again.composition, hmm.composition, atol=1e-5)        # the COMPO line prints five decimals of -log p
            assert again.consensus == hmm.consensus
            occ = hmm.match_occupancy()
            assert occ.shape == (hmm.M + 1,) and occ[0] == 0.0 and np.all((occ[1:] > 0) & (occ[1:] <= 1.0 + 1e-6))
            hmm.validate()
    # copies, equality, pickling (test_hmm.py:48-61, 223-252)
    luxc = models["LuxC"][0]
    for other in (luxc.copy(), copy.copy(luxc), copy.deepcopy(luxc), pickle.loads(pickle.dumps(luxc))):
        assert other == luxc and other is not luxc and other.match_emissions is not luxc.match_emissions
        assert (other.name, other.accession, other.description, other.consensus, other.checksum) == \
               (luxc.name, luxc.accession, luxc.description, luxc.consensus, luxc.checksum)
        assert other.cutoffs.gathering == luxc.cutoffs.gathering and np.array_equal(other._evparam, luxc._evparam)
    assert luxc != models["KR"][0] and luxc != 1 and luxc != plan7.HMM(luxc.alphabet, luxc.M, luxc.name)
    edited = luxc.copy()
    edited.match_emissions[3, 0] += 1e-3
    assert edited != luxc
    with pytest.raises(ValueError, match="match emissions of node 3"):
        edited.validate()
    # counts -> probabilities (test_hmm.py:186-221)
    dna = easel.Alphabet.dna()
    hmm = plan7.HMM(dna, 10, "custom")
    hmm.match_emissions[1:] = 25.0
    hmm.scale(2.0)
    assert np.all(hmm.match_emissions[1:] == 50.0)
    hmm.scale(0.5, exponential=True)                        # node counts 200 -> sqrt(200)
    assert np.allclose(hmm.match_emissions[1:].sum(axis=1), np.sqrt(200.0), rtol=1e-5)
    hmm.renormalize()
    assert np.allclose(hmm.match_emissions.sum(axis=1), 1.0, atol=1e-5) and np.allclose(hmm.insert_emissions.sum(axis=1), 1.0, atol=1e-5)
    t = hmm.transition_probabilities
    assert np.allclose(t[:, 0:3].sum(axis=1), 1.0, atol=1e-5) and t[10, 2] == 0.0 and t[10, 5] == 1.0 and t[10, 6] == 0.0
    assert hmm.composition is None and hmm.consensus is None
    hmm.set_composition()
    assert hmm.composition.shape == (4,) and hmm.composition.sum() == pytest.approx(1.0, abs=1e-5)
    hmm.set_consensus()
    assert len(hmm.consensus) == hmm.M
    seq = easel.TextSequence(sequence="A" * hmm.M).digitize(dna)
    hmm.set_consensus(seq)
    assert hmm.consensus.upper() == "A" * hmm.M
    with pytest.raises(errors.AlphabetMismatch):
        hmm.set_consensus(easel.TextSequence(sequence="Y" * hmm.M).digitize(easel.Alphabet.amino()))
    with pytest.raises(ValueError):
        hmm.set_consensus(easel.TextSequence(sequence="A" * (hmm.M - 1)).digitize(dna))
    hmm.zero()
    assert not hmm.match_emissions.any() and not hmm.transition_probabilities.any() and not hmm.composition.any() and hmm.name == "custom"
    buf = __import__("io").BytesIO()
    plan7.HMM(easel.Alphabet.amino(), 10, "test").write(buf)             # test_hmm.py:170-176: an empty model can be written
    assert len(buf.getvalue()) > 0
    prof = luxc.to_profile(L=200)
    assert prof.M == luxc.M


def test_cutoffs_and_evalue_parameters_are_the_models_own(libp7x, models):
    """reference tests/test_plan7/test_hmm.py:254-340: cutoff pairs read, set, cleared with None or del -- through the view that
    HMM.cutoffs hands out, into the model (a Pipeline with bit_cutoffs sees them) -- and equal after pickling."""
    import pickle
    hmm = models["Thioesterase"][0].copy()
    c = hmm.cutoffs
    for pair in ("gathering", "noise", "trusted"):
        assert getattr(c, pair) is None and getattr(c, pair + "1") is None and getattr(c, pair + "2") is None
        assert not getattr(c, pair + "_available")()
    hmm.cutoffs.gathering = (10.0, 12.0)
    assert hmm.cutoffs.gathering == (10.0, 12.0) and hmm.cutoffs.gathering1 == 10.0 and hmm.cutoffs.gathering2 == 12.0
    assert hmm.cutoffs.gathering_available() and not hmm.cutoffs.noise_available() and not hmm.cutoffs.trusted_available()
    hmm.cutoffs.noise = (8.0, 5.0)
    hmm.cutoffs.trusted = (15.0, 14.0)
    assert (hmm.cutoffs.noise, hmm.cutoffs.noise1, hmm.cutoffs.noise2) == ((8.0, 5.0), 8.0, 5.0)
    assert (hmm.cutoffs.trusted, hmm.cutoffs.trusted1, hmm.cutoffs.trusted2) == ((15.0, 14.0), 15.0, 14.0)
    assert hmm != models["Thioesterase"][0] and "GA    10.00 12.00" in hmm._to_text()
    pf = models["PF02826"][0].copy()
    assert pf.cutoffs.gathering_available() and pf.cutoffs.noise_available() and pf.cutoffs.trusted_available()
    assert pickle.loads(pickle.dumps(pf)).cutoffs == pf.cutoffs and pf.cutoffs != hmm.cutoffs and pf.cutoffs != 3
    pf.cutoffs.gathering = None
    assert pf.cutoffs.gathering is None and pf.cutoffs.gathering1 is None and pf.cutoffs.noise_available()
    del pf.cutoffs.noise
    assert pf.cutoffs.noise is None and pf.cutoffs.noise2 is None and pf.cutoffs.trusted_available()
    pf.cutoffs.trusted = None
    assert not pf.cutoffs.trusted_available() and "Cutoffs gathering=None" in repr(pf.cutoffs)
    ev = hmm.evalue_parameters
    assert ev == models["Thioesterase"][0].evalue_parameters and ev.m_mu == pytest.approx(-10.1820, abs=1e-4)
    hmm.evalue_parameters.m_mu = -9.5
    assert hmm.evalue_parameters.m_mu == -9.5 and hmm.evalue_parameters != models["Thioesterase"][0].evalue_parameters
    hmm.evalue_parameters.f_tau = None
    assert hmm.evalue_parameters.f_tau is None


def test_optimized_profile_cutoffs_and_evalue_parameters_are_read_only():
    """`OptimizedProfile.cutoffs` / `.evalue_parameters` are copies of what the search-ready object holds: a write must not be
    silently dropped -- it raises; the HMM's own views stay writable (reference plan7.pyx:1204-1420, 1760-1849)."""
    hmm = load_hmms("PF02826")[0]
    om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 400)
    with pytest.raises(AttributeError):
        om.cutoffs.gathering = (1.0, 2.0)
    with pytest.raises(AttributeError):
        del om.cutoffs.noise
    with pytest.raises(AttributeError):
        om.evalue_parameters.m_mu = 1.0
    assert om.cutoffs.gathering == hmm.cutoffs.gathering and om.evalue_parameters == hmm.evalue_parameters
    hmm.cutoffs.gathering = (1.0, 2.0)
    assert hmm.cutoffs.gathering == (1.0, 2.0)
