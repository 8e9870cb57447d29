"""Parity of the HIP filter kernels with the CPU oracle, through the C-ABI (p7x_filters_batch / p7x_*_filter).

Integer stages (MSV xJ, Viterbi xC) must be bit-exact; float stages (Forward, bias filter) within the tolerance
stated below (north_star: 'within a stated tolerance on Forward nat scores').
"""
import math

import numpy as np
import pytest

from conftest import synthetic_block
from pyhmmer_amd import easel, plan7

pytestmark = pytest.mark.gpu

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
