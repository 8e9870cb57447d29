"""The device's domain definition against the ORACLE's own (oracle/p7_oracle_dd.c), on the GPU box, through hmmer.hmmsearch
with its defaults -- not against the product's host twin (tests/test_gpu_envelopes.py, test_gpu_ensembles.py do that; VERDICT
r04 item 3).  Reference: p7_domaindef_ByPosteriorHeuristics (include/libhmmer/p7_domaindef.pxd:23-72), SURVEY.md row a13.

The oracle sums every Forward / Backward / null2 float in upstream's striped order (oracle/p7_oracle_dd.c).  The device sums in
its lane chunks and FLAGS what that order cannot be trusted with -- an optimal-accuracy choice on the trace within
cfg.oa_guard of its runner-up, a sampled traceback whose deviate lies within cfg.ens_guard of a threshold -- and the host
stage repeats exactly those envelopes and regions with the same sums in upstream's order (p7x_domaindef.cpp,
forward_full_upstream).  Required: EVERY coordinate of EVERY domain identical (envelope, alignment, model), the same
(nregions, nclustered, noverlaps, nenvelopes), scores and biases within 2e-3 bit (the unflagged envelopes keep the device's
sums; p7_FLogsum's table has steps of 1e-3 nat)."""
import numpy as np
import pytest

from conftest import load_hmms
from test_oracle_domains import _homolog_block
from pyhmmer_amd import easel, hmmer, plan7

pytestmark = pytest.mark.gpu


def _compare(oracle, hmm, hits, sequence_of, bg, rel=False):
    """rel: scores of several hundred bits (long models): the tolerance grows with the score, 2e-3 bit per 100 bits -- a float32
    sum of hundreds of bits carries 1e-5 of representation error on its own (the coordinates stay exact)."""
    op = oracle.OracleProfile(hmm, bg, 400)
    tol = lambda v: 2e-3 * (max(1.0, abs(float(v)) / 100.0) if rel else 1.0)
    stats = {"hits": 0, "single_env": 0, "single_diff": 0, "ens_targets": 0, "ens_same": 0, "domains": 0}
    for h in hits:
        envs, counts = oracle.domains(op, sequence_of(h))
        ours = [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]
        theirs = [tuple(int(v) for v in e[:6]) for e in envs]
        stats["hits"] += 1
        assert h.nregions == counts[0], (h.name, h.nregions, counts)
        if counts[2] == 0:                                   # every region holds one domain
            assert [o[:2] for o in ours] == [t[:2] for t in theirs], (h.name, ours, theirs)          # envelopes: always
            for o, t, e, d in zip(ours, theirs, envs, h.domains):
                stats["single_env"] += 1
                if o != t:
                    stats["single_diff"] += 1
                    stats.setdefault("diffs", []).append((h.name, o, t))
                    continue
                assert abs(d.score - e[9]) <= tol(e[9]) and abs(d.bias - e[10]) <= tol(e[9]), (h.name, d.score, e[9], d.bias, e[10])
                assert abs(d.envelope_score * np.log(2.0) - e[6]) <= 2e-3 * max(1.0, abs(e[6]) / 100.0)
                stats["domains"] += 1
            assert (h.nclustered, h.noverlaps, h.nenvelopes) == (0, 0, counts[1]), h.name
        else:
            stats["ens_targets"] += 1
            if ours != theirs:
                stats.setdefault("diffs", []).append((h.name, ours, theirs))
            if ours == theirs:
                stats["ens_same"] += 1
                assert (h.nregions, h.nclustered, h.noverlaps, h.nenvelopes) == (counts[0], counts[2], counts[4], counts[1]), h.name
                for e, d in zip(envs, h.domains):
                    assert abs(d.score - e[9]) <= tol(e[9]) and abs(d.bias - e[10]) <= tol(e[9]), (h.name, d.score, e[9], d.bias, e[10])
                    stats["domains"] += 1
    return stats


def test_headline_workload_domains_against_the_oracle(oracle):
    """BASELINE configs[1] at full size (KR x 1,000,000 x 300 aa, 1,000 planted): every domain of every hit of
    hmmer.hmmsearch (defaults: device cascade, envelope kernel, device ensembles) against oracle.domains of that target."""
    import bench
    hmm = load_hmms("KR")[0]
    bg = plan7.Background(hmm.alphabet)
    flat, off, ln, planted = bench.make_workload(hmm, 1_000_000, 300, 42)
    db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, off, ln)
    hits = next(iter(hmmer.hmmsearch(hmm, db)))
    assert len(hits) >= 1000
    st = _compare(oracle, hmm, hits, lambda h: np.asarray(flat[off[h.seqidx]:off[h.seqidx] + ln[h.seqidx]], dtype=np.uint8), bg)
    assert st["hits"] >= 1000 and st["single_env"] >= 900, st
    assert st["single_diff"] == 0 and st["ens_same"] == st["ens_targets"], st
    g = hits.guard_counts
    assert 0 < g["oa_redone"] <= st["single_env"] // 10, g          # the guard works, and on a small fraction


@pytest.mark.parametrize("model", ["PF02826", "KR", "LuxC"])
def test_multi_domain_targets_against_the_oracle(oracle, model):
    """Targets with one to three homologous fragments (a third of them multi-domain): regions that hold several domains go
    through the ensembles on the device and in the oracle."""
    hmm = load_hmms(model)[0]
    bg = plan7.Background(hmm.alphabet)
    block = _homolog_block(hmm, 50, 200, seed=21)
    hits = next(iter(hmmer.hmmsearch(hmm, block, E=1e9, domE=1e9, incE=1e9, incdomE=1e9)))
    by_name = {s.name: s for s in block}
    st = _compare(oracle, hmm, hits, lambda h: np.asarray(by_name[h.name].sequence, dtype=np.uint8), bg)
    assert st["hits"] >= 150 and st["ens_targets"] >= 10, st
    assert st["single_diff"] == 0 and st["ens_same"] == st["ens_targets"], st
    g = hits.guard_counts
    assert g["ens_device"] >= 5 and g["ens_redone"] <= max(3, (g["ens_device"] + g["ens_redone"]) // 2), g


def test_region_scan_guard_hands_targets_to_the_host_stage_in_upstream_order(oracle):
    """The device's region scan compares posteriors from its parsers' rows (lane-chunk order) with rt1 / rt2 / rt3; a
    comparison within 2e-5 of its threshold sends the target to the host stage, which forms the parser rows again in
    upstream's order and scans those.  Widened to 0.05 through the test seam the guard takes most targets: the result must
    still be the oracle's, coordinate for coordinate, and equal to the default run's."""
    from pyhmmer_amd import _lib
    hmm = load_hmms("PF02826")[0]
    bg = plan7.Background(hmm.alphabet)
    block = _homolog_block(hmm, 50, 200, seed=21)
    loose = dict(E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
    base = next(iter(hmmer.hmmsearch(hmm, block, **loose)))
    _lib.set_debug_option("region_guard_ppm", 50000)
    try:
        wide = next(iter(hmmer.hmmsearch(hmm, block, **loose)))
    finally:
        _lib.set_debug_option("region_guard_ppm", -1)
    assert wide.guard_counts["region_redone"] >= 50 > base.guard_counts["region_redone"], (wide.guard_counts, base.guard_counts)
    rec = lambda hits: [(h.name, h.nregions, h.nenvelopes, [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]) for h in hits]
    assert rec(wide) == rec(base)
    by_name = {s.name: s for s in block}
    st = _compare(oracle, hmm, wide, lambda h: np.asarray(by_name[h.name].sequence, dtype=np.uint8), bg)
    assert st["single_diff"] == 0 and st["ens_same"] == st["ens_targets"], st


def test_library_shaped_sample_domains_against_the_oracle(oracle):
    """The line's workload in small (BASELINE configs[3]: bench_workloads' Pfam-shaped library and Swiss-Prot-shaped targets): 2,000
    library entries planted into 60,000 targets, and about 40 of them -- spread over the model lengths, with every entry beyond 1,021
    nodes (the eight-lane MSV tiles, 16 lanes per target in the packed Viterbi kernel, 20-32 nodes per lane in the parsers and
    the envelope kernel) -- searched through hmmer.hmmsearch: every domain of every hit against oracle.domains of that target."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench_workloads as bw
    hmms, lib_lengths, templates = bw.make_library(20000, device=0, count=2000)
    bg = plan7.Background(hmms[0].alphabet)
    flat, off, ln, nplanted = bw.make_targets(60_000, 2000, templates, lib_lengths, planted_frac=0.5)
    db = plan7.SequenceDatabase.from_packed(hmms[0].alphabet, flat, off, ln)
    by_len = sorted(range(2000), key=lambda e: hmms[e].M)
    pick = sorted(set(by_len[::80]) | {e for e in range(2000) if hmms[e].M > 1021} | {by_len[-1]})
    assert sum(hmms[e].M > 1021 for e in pick) >= 4 and any(hmms[e].M < 60 for e in pick)
    total = {"hits": 0, "single_env": 0, "single_diff": 0, "ens_targets": 0, "ens_same": 0, "domains": 0}
    for e, hits in zip(pick, hmmer.hmmsearch([hmms[e] for e in pick], db)):
        st = _compare(oracle, hmms[e], hits, lambda h: np.asarray(flat[off[h.seqidx]:off[h.seqidx] + ln[h.seqidx]], dtype=np.uint8), bg, rel=True)
        assert st["single_diff"] == 0 and st["ens_same"] == st["ens_targets"], (e, hmms[e].M, st)
        for k in total:
            total[k] += st[k]
    print("library-shaped sample:", len(pick), "profiles, M", min(hmms[e].M for e in pick), "...", max(hmms[e].M for e in pick), total)
    assert total["hits"] >= 300 and total["domains"] >= 300, total
