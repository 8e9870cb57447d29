"""The CPU oracle pinned against the reference's own fixtures.  No GPU."""
import numpy as np
import pytest

import h3_reader
from conftest import GOLDEN, golden_table
from pyhmmer_amd import plan7

# stage counts of the cascade on the 2,100-protein fixture proteome
STAGE_COUNTS = {
    "2-Hacid_dh_C": (130, 130, 48, 22), "Thioesterase": (27, 27, 7, 1),
    "Stand_Alone_Lasso_RRE": (96, 75, 5, 1), "Thiopeptide_F_RRE": (61, 59, 1, 0), "PqqD_RRE": (106, 88, 8, 2),
    "Proteusin_Epimerase_RRE": (78, 62, 5, 1), "Thurincin_rSAM_RRE": (209, 51, 7, 1), "Thuricin_rSAM_RRE": (294, 84, 10, 1),
    "Other_Sactipeptide_rSAM_RRE": (181, 71, 10, 1), "Ranthipeptide_rSAM_RRE": (222, 109, 8, 2),
    "Trifolitoxin_RRE": (47, 39, 1, 0), "Thiaglutamate_B_RRE": (42, 42, 2, 1),
}


@pytest.mark.parametrize("name", ["PF02826", "Thioesterase", "RREFam"])
def test_oracle_tables_bit_exact_vs_pressed(name, models, oracle):
    """p7_ProfileConfig + p7_oprofile_Convert restatement == files pressed by real HMMER 3.3.1."""
    f = h3_reader.read_h3f(GOLDEN / "db" / f"{name}.hmm.h3f")
    p = h3_reader.read_h3p(GOLDEN / "db" / f"{name}.hmm.h3p")
    assert len(f) == len(p) == len(models[name])
    for hmm, ff, pp in zip(models[name], f, p):
        op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
        assert np.array_equal(op.arr("rbv"), ff["rbv"])
        assert np.array_equal(op.arr("sbv"), ff["sbv"])
        assert np.array_equal(op.arr("rwv"), pp["rwv"])
        assert np.array_equal(op.arr("twv"), pp["twv"])
        assert np.array_equal(op.arr("rfv").view(np.uint32), pp["rfv"].view(np.uint32))   # Easel's vector expf, bit for bit
        assert np.array_equal(op.arr("tfv").view(np.uint32), pp["tfv"].view(np.uint32))
        o = op.p
        assert (o.tbm_b, o.tec_b, o.tjb_b, o.base_b, o.bias_b) == (ff["tbm"], ff["tec"], ff["tjb"], ff["base"], ff["bias"])
        assert o.ddbound_w == pp["ddbound_w"] and o.base_w == pp["base_w"]
        assert [[o.xw[i][j] for j in range(2)] for i in range(4)] == pp["xw"].tolist()


@pytest.mark.parametrize("name,table", [("PF02826", "PF02826.tbl"), ("Thioesterase", None), ("RREFam", "RREFam.tbl")])
def test_oracle_cascade_survivors_are_the_golden_hits(name, table, models, oracle, proteome):
    pk = proteome.packed()
    for hmm in models[name]:
        op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
        recs, ctr = op.cascade_block(pk)
        surv = sorted(proteome[i].name for i in range(len(proteome)) if recs[i].stage == 4)
        if table is None:
            gold = ["938293.PRJEB85.HG003687_113"]          # reference tests/test_hmmer.py:80-85
        else:
            gold = sorted(r[0] for r in golden_table(table, hmm.name))
        assert surv == gold
        assert (ctr.n_past_msv, ctr.n_past_bias, ctr.n_past_vit, ctr.n_past_fwd) == STAGE_COUNTS[hmm.name]


def test_oracle_pre_score_matches_table(models, oracle, proteome):
    """score + bias of PF02826.tbl pins (fwd - null1)/ln2 to 0.1 bit."""
    hmm = models["PF02826"][0]
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    by_name = {s.name: s for s in proteome}
    for row in golden_table("PF02826.tbl"):
        s = by_name[row[0]]
        st, fsc = op.fwd(s.sequence)
        pre = (fsc - oracle.lib().p7o_null1(len(s))) / np.log(2)
        assert abs(pre - (float(row[5]) + float(row[6]))) < 0.11


def test_oracle_sse_equals_scalar_twin(models, oracle, proteome):
    hmm = models["PF02826"][0]
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    for s in proteome[::13]:
        assert op.msv(s.sequence) == op.msv(s.sequence, scalar=True)
        assert op.vit(s.sequence) == op.vit(s.sequence, scalar=True)


def test_oracle_forward_equals_backward(models, oracle, proteome):
    hmm = models["Thioesterase"][0]
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    for s in proteome[:40:4]:
        st, fsc = op.fwd(s.sequence)
        st2, bsc, fx, bx = op.bck(s.sequence)
        assert st == 0 and st2 == 0
        assert abs(fsc - bsc) < 1e-3 * max(1.0, abs(fsc))


def test_oracle_null1_doctest_value(oracle):
    """reference plan7.pyx:579-583: bg.null1 of a length-70 sample with the default p1 = 350/351."""
    L, p1 = 70, np.float32(350) / np.float32(351)
    v = L * np.log(np.float64(p1)) + np.log(1.0 - np.float64(p1))
    assert abs(v - (-6.0605)) < 5e-3
    # and the oracle's own null1 (p1 = L/(L+1)) for a few lengths
    for L in (1, 50, 400, 100000):
        ref = L * np.log(L / (L + 1.0)) + np.log(1.0 / (L + 1.0))
        assert abs(oracle.lib().p7o_null1(L) - ref) < 2e-3 * abs(ref)


@pytest.mark.parametrize("name,T,P", [("KR", 8, 17), ("PF02826", 8, 12), ("PF02826", 16, 11)])
def test_striped_packed_viterbi_algorithm_equals_oracle(name, T, P, models, oracle, proteome):
    """The node order, stripe shift and closure rule of p7x_vitpk.hip, restated in numpy (tests/vit_striped_emu.py),
    give the oracle's Viterbi xC: the kernel's algorithm is checked here without a GPU, its code on the GPU box."""
    import numpy as np
    import vit_striped_emu
    from pyhmmer_amd import plan7
    hmm = models[name][0]
    op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
    rng = np.random.default_rng(11)
    passes = []
    for si in rng.integers(0, len(proteome), 4):
        seq = np.array(proteome[int(si)].sequence, dtype=np.uint8)[:120]
        st, sc, xc = op.vit(seq)
        assert vit_striped_emu.vit_striped(op, seq, T, P, passes) == xc
    assert passes and max(passes) <= 2 * T
