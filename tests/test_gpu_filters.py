"""Parity of the HIP filter kernels with the CPU oracle, through the C-ABI (p7x_filters_batch / p7x_*_filter).

Integer stages (MSV xJ, Viterbi xC) must be bit-exact; float stages (Forward, bias filter) within the tolerance
stated below (north_star: 'within a stated tolerance on Forward nat scores').
"""
import math

import numpy as np
import pytest

from conftest import load_hmms, random_hmm, synthetic_block
from pyhmmer_amd import easel, plan7

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

FWD_TOL_NATS = 2e-3      # |fwd_gpu - fwd_oracle| in nats (float32 sums in a different association order)
BIAS_TOL_NATS = 1e-3


def _oracle_scores(op, block, want=("msv", "vit", "fwd", "bias")):
    out = {k: [] for k in want}
    for s in block:
        if "msv" in want:
            out["msv"].append(op.msv(s.sequence)[2])
        if "vit" in want:
            out["vit"].append(op.vit(s.sequence)[2])
        if "fwd" in want:
            out["fwd"].append(op.fwd(s.sequence)[1])
        if "bias" in want:
            out["bias"].append(op.bias(s.sequence))
    return {k: np.array(v) for k, v in out.items()}


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam", "KR", "LuxC"])
def test_msv_bit_exact_on_fixture_proteome(name, models, oracle, proteome):
    db = plan7.SequenceDatabase(proteome)
    for hmm in models[name][:3]:
        bg = plan7.Background(hmm.alphabet)
        om = plan7.OptimizedProfile(hmm, bg, 400)
        got = db.filters(om, msv=True)["xJ"]
        want = oracle.OracleProfile(hmm, bg, 400).msv_block(proteome.packed())
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{hmm.name}: {bad.size} targets differ, first {bad[:5]} got {got[bad[:5]]} want {want[bad[:5]]}"


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam"])
def test_viterbi_forward_bias_parity_on_fixture_subset(name, models, oracle, proteome):
    sub = easel.DigitalSequenceBlock(proteome.alphabet, list(proteome[::7]))
    db = plan7.SequenceDatabase(sub)
    for hmm in models[name][:2]:
        bg = plan7.Background(hmm.alphabet)
        om = plan7.OptimizedProfile(hmm, bg, 400)
        got = db.filters(om, msv=False, viterbi=True, forward=True, bias=True)
        want = _oracle_scores(oracle.OracleProfile(hmm, bg, 400), sub, want=("vit", "fwd", "bias"))
        assert np.array_equal(got["xC"], want["vit"]), np.nonzero(got["xC"] != want["vit"])[0][:10]
        assert np.max(np.abs(got["fwd"] - want["fwd"])) < FWD_TOL_NATS
        assert np.max(np.abs(got["filtersc"] - want["bias"])) < BIAS_TOL_NATS


def test_ragged_lengths_and_tile_edges(models, oracle):
    """Lengths around the 16-residue tile blocks and the 64-sequence groups, incl. L = 1."""
    hmm = models["PF02826"][0]
    bg = plan7.Background(hmm.alphabet)
    lengths = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 300, 1000, 4561] + list(range(40, 140))
    blk = synthetic_block(len(lengths), 0, seed=7, lengths=lengths)
    db = plan7.SequenceDatabase(blk)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    op = oracle.OracleProfile(hmm, bg, 400)
    got = db.filters(om, msv=True, viterbi=True, forward=True)
    want = _oracle_scores(op, blk, want=("msv", "vit", "fwd"))
    assert np.array_equal(got["xJ"], want["msv"])
    assert np.array_equal(got["xC"], want["vit"])
    assert np.max(np.abs(got["fwd"] - want["fwd"])) < FWD_TOL_NATS


def test_empty_targets_and_empty_block(models):
    hmm = models["PF02826"][0]
    abc = hmm.alphabet
    om = plan7.OptimizedProfile(hmm, plan7.Background(abc), 100)
    blk = synthetic_block(5, 0, seed=3, lengths=[0, 10, 0, 200, 0])
    got = plan7.SequenceDatabase(blk).filters(om, msv=True)["xJ"]
    assert got.shape == (5,) and got[0] == 0 and got[2] == 0 and got[4] == 0
    empty = easel.DigitalSequenceBlock(abc, [])
    assert plan7.SequenceDatabase(empty).filters(om, msv=True)["xJ"].shape == (0,)
    hits = plan7.Pipeline(abc).search_hmm(hmm, empty)
    assert len(hits) == 0 and hits.searched_sequences == 0


def test_msv_overflow_is_reported_as_infinity(models, oracle):
    """A target made of the model's own consensus overflows the 8-bit MSV score: eslERANGE / +inf."""
    hmm = models["PF02826"][0]
    abc = hmm.alphabet
    bg = plan7.Background(abc)
    cons = np.array([int(np.argmax(hmm.match_emissions[k])) for k in range(1, hmm.M + 1)], dtype=np.uint8)
    seq = easel.DigitalSequence(abc, name="consensus", sequence=np.tile(cons, 2))
    om = plan7.OptimizedProfile(hmm, bg, 400)
    op = oracle.OracleProfile(hmm, bg, 400)
    st, sc, xj = op.msv(seq.sequence)
    assert st == 16 and xj == -1
    assert math.isinf(om.msv_filter(seq))
    got = plan7.SequenceDatabase(easel.DigitalSequenceBlock(abc, [seq])).filters(om, msv=True, viterbi=True)
    assert got["xJ"][0] == -1
    assert got["xC"][0] == op.vit(seq.sequence)[2]


def test_single_sequence_entry_points(models, oracle, proteome):
    hmm = models["Thioesterase"][0]
    bg = plan7.Background(hmm.alphabet)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    op = oracle.OracleProfile(hmm, bg, 400)
    for s in proteome[5:8]:
        assert om.msv_filter(s) == pytest.approx(op.msv(s.sequence)[1], abs=1e-6)
        assert om.ssv_filter(s) == pytest.approx(op.msv(s.sequence)[1], abs=1e-6)
        assert om.viterbi_filter(s) == pytest.approx(op.vit(s.sequence)[1], abs=1e-6)
        assert om.forward_parser(s) == pytest.approx(op.fwd(s.sequence)[1], abs=FWD_TOL_NATS)
        assert om.backward_parser(s) == pytest.approx(op.bck(s.sequence)[1], abs=5e-3)


def test_degenerate_and_special_residues(models, oracle):
    """X, B, Z, J, U, O and '*' codes go through the emission tables like any other row."""
    hmm = models["PF02826"][0]
    abc = hmm.alphabet
    bg = plan7.Background(abc)
    rng = np.random.default_rng(11)
    seqs = []
    for t in range(70):
        x = rng.integers(0, 20, size=200).astype(np.uint8)
        x[rng.integers(0, 200, size=12)] = rng.choice([21, 22, 23, 24, 25, 26, 27], size=12)
        seqs.append(easel.DigitalSequence(abc, name=f"d{t}", sequence=x))
    blk = easel.DigitalSequenceBlock(abc, seqs)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    got = plan7.SequenceDatabase(blk).filters(om, msv=True, viterbi=True, forward=True, bias=True)
    want = _oracle_scores(oracle.OracleProfile(hmm, bg, 400), blk)
    assert np.array_equal(got["xJ"], want["msv"]) and np.array_equal(got["xC"], want["vit"])
    assert np.max(np.abs(got["fwd"] - want["fwd"])) < FWD_TOL_NATS
    assert np.max(np.abs(got["filtersc"] - want["bias"])) < BIAS_TOL_NATS


def test_msv_bit_exact_on_large_synthetic_block(models, oracle):
    """64k synthetic 300-aa targets (BASELINE config-2 shape, scaled to oracle speed): checksum of all xJ."""
    hmm = models["KR"][0]
    bg = plan7.Background(hmm.alphabet)
    blk = synthetic_block(20000, 300, seed=42)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    got = plan7.SequenceDatabase(blk).filters(om, msv=True)["xJ"]
    want = oracle.OracleProfile(hmm, bg, 400).msv_block(blk.packed())
    assert np.array_equal(got, want)


def test_msv_floor_ambiguous_targets_fall_back_to_exact_kernel(models, oracle):
    """Targets whose every row maximum stays below the begin score (all-'*', all-X, poly-W ...) are exactly the
    ones the fast MSV kernel cannot resolve (xJ == F0): their groups must be redone by the exact kernel."""
    hmm = models["PF02826"][0]
    abc = hmm.alphabet
    bg = plan7.Background(abc)
    rng = np.random.default_rng(5)
    seqs = []
    for t, code in enumerate([27, 26, 18, 1, 27, 26]):          # '*', 'X', 'W', 'C'
        seqs.append(easel.DigitalSequence(abc, name=f"deg{t}", sequence=np.full(40 + 13 * t, code, dtype=np.uint8)))
    for t in range(70):                                           # normal neighbours in the same 64-target groups
        seqs.append(easel.DigitalSequence(abc, name=f"n{t}", sequence=rng.integers(0, 20, size=60 + t).astype(np.uint8)))
    seqs.append(easel.DigitalSequence(abc, name="mix", sequence=np.array([27] * 30 + [0, 5, 9] + [27] * 30, dtype=np.uint8)))
    blk = easel.DigitalSequenceBlock(abc, seqs)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    got = plan7.SequenceDatabase(blk).filters(om, msv=True)["xJ"]
    want = oracle.OracleProfile(hmm, bg, 400).msv_block(blk.packed())
    assert np.array_equal(got, want), (got[:8], want[:8])


def _model_block(hmm, nrand, nhom, seed):
    """Random background targets of ragged length plus a few sequences emitted from the model itself."""
    import bench
    abc = hmm.alphabet
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 420, size=nrand)
    blk = synthetic_block(nrand, 0, seed=seed, alphabet=abc, lengths=lens)
    seqs = list(blk)
    t = hmm.transition_probabilities.astype(np.float64)
    mat, ins = hmm.match_emissions.astype(np.float64), hmm.insert_emissions.astype(np.float64)
    cmat = np.cumsum(mat / np.maximum(mat.sum(axis=1, keepdims=True), 1e-30), axis=1)
    cins = np.cumsum(ins / np.maximum(ins.sum(axis=1, keepdims=True), 1e-30), axis=1)
    ct = np.zeros((hmm.M + 1, 4))
    s3 = np.maximum(t[:, 0:3].sum(axis=1), 1e-30)
    ct[:, 0], ct[:, 1] = t[:, 0] / s3, (t[:, 0] + t[:, 1]) / s3
    ct[:, 2] = t[:, 3] / np.maximum(t[:, 3] + t[:, 4], 1e-30)
    ct[:, 3] = t[:, 5] / np.maximum(t[:, 5] + t[:, 6], 1e-30)
    for h in range(nhom):
        dom = bench.emit_from_model(hmm, rng, (cmat, cins, ct))
        lo = int(rng.integers(0, max(1, len(dom) // 2)))
        hi = int(rng.integers(lo + 1, len(dom) + 1))
        flank = rng.integers(0, abc.K, size=int(rng.integers(0, 50))).astype(np.uint8)
        seqs.append(easel.DigitalSequence(abc, name=f"hom{h}", sequence=np.concatenate([flank, dom[lo:hi], flank])))
    return easel.DigitalSequenceBlock(abc, seqs)


# one model length per register-count instantiation of the MSV kernels (R = M/2+1 rounded up to the R list) and per
# nodes-per-lane instantiation C of the wavefront kernels (C = ceil(M/64) rounded up to the C list)
# K lanes per target: one lane up to 224 registers, two lanes (8-register steps), four lanes, and eight lanes with uniform
# rows (one table, one register alignment: register j = cells 2j+1, 2j+2) up to 2,048 nodes (p7x_msv.hip: msv_pick)
_R_LIST = list(range(8, 161, 4)) + [176, 192, 208, 224]
_R_LIST2 = list(range(120, 225, 8))
_R_LIST4 = [120, 128]
_R_LIST8 = list(range(64, 129, 8))
SWEEP_M = ([1, 2, 3] + [m for r in _R_LIST for m in (2 * (r - 1) - 1, 2 * (r - 1))]       # largest odd and even M per R
           + [m for r in _R_LIST2 for m in (2 * (2 * r - 1) - 1, 2 * (2 * r - 1))]
           + [m for r in _R_LIST4 for m in (2 * (4 * r - 1) - 1, 2 * (4 * r - 1))]
           + [m for r in _R_LIST8 if r > 64 for m in (16 * r - 1, 16 * r)]
           + [447, 448, 449, 894, 895, 1022, 1023, 1024, 1025, 2049])                       # the seams between the families


@pytest.mark.parametrize("M", SWEEP_M)
def test_every_msv_kernel_instantiation_bit_exact(M, oracle):
    hmm = random_hmm(M, seed=1000 + M)
    bg = plan7.Background(hmm.alphabet)
    blk = _model_block(hmm, 400, 12, seed=M)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    op = oracle.OracleProfile(hmm, bg, 400)
    for rep in range(2):          # twice: hazards in hand-scheduled LDS traffic are timing dependent
        got = plan7.SequenceDatabase(blk).filters(om, msv=True)["xJ"]
        assert np.array_equal(got, op.msv_block(blk.packed())), f"M={M} rep={rep}"


@pytest.mark.parametrize("M", [5, 60, 262, 263, 445, 446, 700, 893, 1000, 1021, 1300, 2048])
def test_integer_flavour_of_the_fast_msv_kernel_bit_exact(M, oracle):
    """The lane-per-target MSV kernel has two flavours of its floored representation: binary16 cells (the default since
    round 5: v_pk_add_f16 clamp + v_pk_maximum3_f16) and int16 cells (v_pk_add_i16 clamp + v_pk_max_i16).  Everything else
    in this file runs the first; this sends one model per lanes-per-target family through the second (seam "msv_f16")."""
    from pyhmmer_amd import _lib
    hmm = random_hmm(M, seed=1000 + M)
    bg = plan7.Background(hmm.alphabet)
    blk = _model_block(hmm, 400, 12, seed=M)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    want = oracle.OracleProfile(hmm, bg, 400).msv_block(blk.packed())
    _lib.set_debug_option("small_block", 0)
    try:
        for flavour in (0, 1):
            _lib.set_debug_option("msv_f16", flavour)
            got = plan7.SequenceDatabase(blk).filters(om, msv=True)["xJ"]
            assert np.array_equal(got, want), f"M={M} flavour={'half' if flavour else 'int16'}"
    finally:
        _lib.set_debug_option("msv_f16", -1)


# wave-per-target kernels: one M per nodes-per-lane count C; packed Viterbi kernel (M <= 640): the largest M of every
# (lanes per target T, register pairs per lane P) instantiation, i.e. 16 P for T = 8 and 32 P for T = 16, and one below
_PK_M = sorted({16 * p - d for p in (2, 4, 6, 8, 10, 12, 14, 15, 16, 17, 18, 19, 20) for d in (0, 1)} |
               {32 * p - d for p in range(11, 21) for d in (0, 1)})


@pytest.mark.parametrize("M", sorted(set([1, 3, 64, 65, 128, 150, 192, 256, 300, 320, 384, 479, 500, 512, 640, 641, 700, 768, 1000, 1024, 1025,
                                          1280, 1500, 1536, 2000, 2048,
                                          2049, 2304, 2305, 2560, 2561, 3000, 3072, 4096, 4097, 5000, 6144, 8192] + _PK_M)))   # M > 2048: rolled loops, tables through L2
def test_every_wavefront_kernel_instantiation_vs_oracle(M, oracle):
    hmm = random_hmm(M, seed=2000 + M)
    bg = plan7.Background(hmm.alphabet)
    blk = _model_block(hmm, 90, 6, seed=M)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    got = plan7.SequenceDatabase(blk).filters(om, msv=True, viterbi=True, forward=True)
    want = _oracle_scores(oracle.OracleProfile(hmm, bg, 400), blk, want=("msv", "vit", "fwd"))
    assert np.array_equal(got["xJ"], want["msv"]), f"M={M}"      # M > 478: the wave-per-target MSV kernel
    assert np.array_equal(got["xC"], want["vit"]), f"M={M}"
    ok = np.isfinite(want["fwd"])
    assert np.array_equal(np.isfinite(got["fwd"]), ok)
    # float32 scores of several hundred nats carry ~1e-4 of representation error on their own
    assert np.all(np.abs(got["fwd"][ok] - want["fwd"][ok]) < FWD_TOL_NATS + 1e-5 * np.abs(want["fwd"][ok]))


@pytest.mark.parametrize("M", [1022, 1281, 1537, 2047, 2100, 2500])      # (beyond 2,048: 36 / 40 nodes per lane, Kp + 1 table rows)
def test_packed_wave_msv_for_long_models_on_a_ragged_block(M, oracle):
    """Models beyond the lane kernels (M > 1021) run msv_wavepk_kernel: packed pairs, sixteen wavefronts per block, eight
    rows per reduction with the begin score held and the block repeated row by row where a hit moved it.  1,500 random
    targets of 1 ... 420 residues (row tails of every length, several blocks per launch) plus 40 fragments emitted
    from the model (begin score moves), every xJ against the oracle, twice."""
    hmm = random_hmm(M, seed=3000 + M)
    bg = plan7.Background(hmm.alphabet)
    blk = _model_block(hmm, 1500, 40, seed=M)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    want = oracle.OracleProfile(hmm, bg, 400).msv_block(blk.packed())
    assert int((want > 0).sum()) > 40                      # -1 marks overflow (an infinite score): the fragments do overflow
    db = plan7.SequenceDatabase(blk)
    for rep in range(2):
        got = db.filters(om, msv=True)["xJ"]
        assert np.array_equal(got, want), f"M={M} rep={rep}"


def test_msv_baseline_config2_full_size_vs_oracle(oracle):
    """BASELINE.json configs[1] at full size: 10^6 synthetic 300-residue targets, every xJ against the oracle."""
    import bench
    from types import SimpleNamespace
    hmm = load_hmms("KR")[0]
    bg = plan7.Background(hmm.alphabet)
    flat, off, ln, planted = bench.make_workload(hmm, 1_000_000, 300, 42)
    db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, off, ln)
    om = plan7.OptimizedProfile(hmm, bg, 300)
    want = oracle.OracleProfile(hmm, bg, 300).msv_block(SimpleNamespace(dsq=flat, offsets=off, lengths=ln, n=len(ln)))
    for rep in range(2):
        got = db.filters(om, msv=True)["xJ"]
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (rep, bad[:10], got[bad[:10]], want[bad[:10]])


@pytest.mark.parametrize("kind", ["dna", "rna"])
def test_nucleotide_alphabets_through_every_filter(kind, oracle):
    """The standard pipeline is alphabet-agnostic (Kp = 18 residue rows instead of 29)."""
    abc = easel.Alphabet.dna() if kind == "dna" else easel.Alphabet.rna()
    hmm = random_hmm(150, seed=77, alphabet=abc)
    bg = plan7.Background(abc)
    blk = _model_block(hmm, 200, 10, seed=5)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    got = plan7.SequenceDatabase(blk).filters(om, msv=True, viterbi=True, forward=True, bias=True)
    want = _oracle_scores(oracle.OracleProfile(hmm, bg, 400), blk)
    assert np.array_equal(got["xJ"], want["msv"]) and np.array_equal(got["xC"], want["vit"])
    ok = np.isfinite(want["fwd"])
    assert np.all(np.abs(got["fwd"][ok] - want["fwd"][ok]) < FWD_TOL_NATS + 1e-5 * np.abs(want["fwd"][ok]))
    assert np.max(np.abs(got["filtersc"] - want["bias"])) < BIAS_TOL_NATS
    hits = plan7.Pipeline(abc, E=1e3).search_hmm(hmm, blk)
    assert len(hits) >= 4 and all(h.domains[0].alignment.target_sequence for h in hits)


def test_bias_filter_logarithm_equals_the_library_logarithm_for_every_float():
    """bias_kernel takes (float) log((double) x) at every residue through log_of_float (a 128-entry table and five series
    terms in double) instead of the library call the reference makes in esl_hmm_Forward.  The arguments are sums of two
    probability x odds products, of order one: EVERY float of [2^-7, 2^7) -- 14 binades, 117,440,512 values -- must give
    the same float as the host's log(); zero, denormals, infinity and NaN take the library path."""
    import ctypes as C
    from pyhmmer_amd import _lib
    bad = 0
    for e in range(-7, 7):
        bits = np.arange(1 << 23, dtype=np.uint32) | np.uint32((127 + e) << 23)
        x = bits.view(np.float32)
        out = np.empty_like(x)
        assert _lib.lib().p7x_debug_log_of_float(0, x.ctypes.data, out.ctypes.data, x.size) == 0, _lib.last_error()
        want = np.log(x.astype(np.float64)).astype(np.float32)
        bad += int(np.count_nonzero(out != want))
    assert bad == 0, bad
    x = np.array([0.0, 1e-42, np.inf, 1.0, 2.0], dtype=np.float32)
    out = np.empty_like(x)
    assert _lib.lib().p7x_debug_log_of_float(0, x.ctypes.data, out.ctypes.data, x.size) == 0
    assert out[0] == -np.inf and out[2] == np.inf and out[3] == 0.0 and out[4] == np.float32(np.log(2.0)) and abs(out[1] - np.log(1e-42)) < 1e-3


@pytest.mark.parametrize("M", [20, 33, 64, 65, 100, 128, 129, 160, 192, 193, 230, 256, 257, 262, 320, 321, 384])
def test_grouped_forward_parser_against_the_wave_per_target_kernel_and_the_oracle(oracle, M):
    """p7x_fwdpk.hip (T = 16 or 32 lanes per target, the first Forward pass against large blocks; option fwd_grouped) for every (T, C)
    instantiation and its boundary lengths: 20,000 ragged targets with 300 homologs; the same stage counts, the same hits
    and scores within the parser's stated tolerance as with fwd_kernel (option fwd_grouped = 0), and the oracle's Forward
    scores (upstream's striped order) within 2e-3 nat for every target that reaches the parser."""
    from pyhmmer_amd import _lib
    hmm = random_hmm(M, seed=7000 + M)
    blk = _model_block(hmm, 20_000, 300, seed=M)
    db = plan7.SequenceDatabase(blk)
    assert len(blk) > 256 * 64
    pli = dict(E=1e3, domE=1e3)
    old = plan7.Pipeline(hmm.alphabet, **pli).search_hmm(hmm, db)
    _lib.set_debug_option("fwd_grouped", 1)
    try:
        new = plan7.Pipeline(hmm.alphabet, **pli).search_hmm(hmm, db)
    finally:
        _lib.set_debug_option("fwd_grouped", -1)
    assert new.stage_counts == old.stage_counts and new.stage_counts["vit"] >= 250
    assert [h.name for h in new] == [h.name for h in old]
    for a, b in zip(new, old):
        assert abs(a.pre_score - b.pre_score) <= 4e-3 and abs(a.score - b.score) <= 4e-3, (a.name, a.pre_score, b.pre_score)
    # the oracle's parser on the hits: pre_score = (fwd - null1) / ln 2
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    by_name = {s.name: s for s in blk}
    n = match = 0
    for h in list(new)[:120]:
        seq = np.asarray(by_name[h.name].sequence, dtype=np.uint8)
        st, fsc = op.fwd(seq)
        L = len(seq)
        null1 = L * np.log(L / (L + 1.0)) + np.log(1.0 / (L + 1.0))
        n += 1
        match += abs(h.pre_score - (fsc - null1) / np.log(2.0)) <= 4e-3      # (not when the sum of the domains replaced the Forward score: p7_pipeline.c)
    assert n >= 50 and match >= 0.6 * n, (match, n)
