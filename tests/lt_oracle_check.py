"""The long-target pipeline's tail checked from OUTSIDE the product: oracle/p7_oracle_lt.c restates window merging, the MSV /
bias tests, the long-target Viterbi scan, the Forward test and the scoring rule of an envelope apart from the product's
p7x_longtarget.inc.hpp; this module walks the (block, strand) units of a search as the reference's loop does
(plan7.pyx:7541-7664) and compares a product hit list with what the oracle lets through."""
import numpy as np

import oracle_lib
from host_pipeline import DNA_COMP, blocks_of
from pyhmmer_amd import plan7

LT_SCORE_TOL_BITS = 1e-3      # product score vs the oracle's scoring rule applied to the hit's own envelope / alignment coordinates
LT_ENV_TOL_NATS = 5e-3        # the envelope's Forward scores (own odds / composition-adjusted odds), float sums in another order


def synthetic_chromosome(hmm, L, seed, block_length, max_length, long_copies=True):
    """i.i.d. ACGT with everything the tail has branches for: full and partial copies of the model's consensus on both strands
    (20 % mutated), hundreds of short fragments (40 - 95 nt: envelopes below 100 residues, the short-window branch of the
    background mix), low-complexity stretches (homopolymers, dinucleotide repeats, AT-rich runs: the bias filter's work), and
    copies laid across the seams between blocks."""
    rng = np.random.default_rng(seed)
    seq = rng.integers(0, 4, size=L, dtype=np.uint8)
    cons = np.argmax(hmm.match_emissions[1:], axis=1).astype(np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)

    def plant(pos, a, n, rate, rev):
        seg = cons[a:a + n].copy()
        mut = rng.random(len(seg)) < rate
        seg[mut] = rng.integers(0, 4, size=int(mut.sum()))
        if pos < 0 or pos + len(seg) > L:
            return
        seq[pos:pos + len(seg)] = comp[seg[::-1]] if rev else seg

    for _ in range(max(6, L // 200_000)):                       # low complexity first (copies may land inside)
        n = int(rng.integers(200, 2500)); pos = int(rng.integers(0, L - n)); kind = int(rng.integers(0, 3))
        if kind == 0:
            seq[pos:pos + n] = rng.integers(0, 4)
        elif kind == 1:
            seq[pos:pos + n] = np.resize(rng.integers(0, 4, size=2).astype(np.uint8), n)
        else:
            seq[pos:pos + n] = rng.choice(np.array([0, 3], dtype=np.uint8), size=n)
    if long_copies:
        for c in range(max(8, L // 35_000)):
            a = int(rng.integers(0, max(1, hmm.M - 200))); n = min(int(rng.integers(150, 1200)), hmm.M - a)
            plant(int(rng.integers(0, L - n)), a, n, 0.2, c % 2 == 1)
    for c in range(max(40, L // 1_400)):                         # short fragments
        n = min(int(rng.integers(40, 96)), hmm.M); a = int(rng.integers(0, hmm.M - n + 1))
        plant(int(rng.integers(0, L - n)), a, n, float(rng.choice([0.0, 0.05, 0.1])), c % 2 == 1)
    step = block_length - max_length                             # seams: block k starts at k * step
    for k in range(1, L // step + 1):
        n = min(600, hmm.M); a = int(rng.integers(0, hmm.M - n + 1))
        plant(k * step - n // 2 + int(rng.integers(-10, 10)), a, n, 0.15, k % 2 == 1)
        plant(k * step + max_length - n // 2, a, n, 0.15, k % 2 == 0)
    return seq


def oracle_windows(pipeline, hmm, seq):
    """Per (block start, strand): the oracle's Forward-passing windows; and the summed stage counts."""
    op = oracle_lib.OracleProfile(hmm, pipeline.background, 400)
    max_length = pipeline.window_length or hmm.max_length
    total = np.zeros(8, dtype=np.uint64)
    units = {}
    for (i, n) in blocks_of(len(seq), pipeline.block_length, max_length):
        for strand in (0, 1):
            blk = seq[i:i + n] if strand == 0 else DNA_COMP[seq[i:i + n][::-1]]
            win, counts = oracle_lib.lt_block(op, blk, max_length, pipeline.F1, pipeline.F2, pipeline.F3, pipeline.B1, pipeline.B2,
                                              pipeline.B3, pipeline.bias_filter)
            units[(i, n, strand)] = win
            total += counts
    return op, max_length, units, total


def check_hits_against_oracle(pipeline, hmm, seq, hits, min_windows=0, min_short=0, want_short_windows=False, oracle=None):
    op, max_length, units, total = oracle or oracle_windows(pipeline, hmm, seq)
    sc = hits.stage_counts
    assert (sc["msv"], sc["bias"], sc["vit"], sc["fwd"]) == tuple(int(v) for v in total[:4]), (sc, total)
    assert int(total[2]) >= min_windows                          # windows the Forward parser ran on
    nshort = nshortwin = 0
    for h in hits:
        d = h.domains[0]
        a = d.alignment
        # the strand column compares the alignment's ends and reads "-" when they are one residue (as upstream prints it):
        # the envelope's ends tell
        rev = d.env_from > d.env_to if d.env_from != d.env_to else d.strand == "-"
        lo, hi = min(d.env_from, d.env_to), max(d.env_from, d.env_to)
        inside, wlens = False, []
        for (i, n, strand), win in units.items():
            if strand != int(rev) or len(win) == 0:
                continue
            # block coordinates of the envelope on this strand
            b_lo, b_hi = (lo - i, hi - i) if not rev else (i + n - hi + 1, i + n - lo + 1)
            if b_lo < 1 or b_hi > n:
                continue
            holds = (win[:, 0] <= b_lo) & (win[:, 0] + win[:, 1] - 1 >= b_hi)
            if np.any(holds):
                inside = True
                wlens += [int(v) for v in win[holds, 1]]
        assert inside, (h.name, d.env_from, d.env_to, d.strand)
        env_len = hi - lo + 1
        ali_len = abs(a.target_to - a.target_from) + 1
        nshort += env_len < 100
        ln2 = float(np.log(2.0))                                  # the Python properties are in bits, the rule works in nats
        want, bias_bits, _ = oracle_lib.lt_domain_score(op, max_length, env_len, ali_len, d.envelope_score * ln2, d.correction * ln2, pipeline.null2)
        assert abs(want - h.score) <= LT_SCORE_TOL_BITS, (h.name, d.env_from, d.env_to, want, h.score)
        assert abs(bias_bits - d.bias) <= LT_SCORE_TOL_BITS, (h.name, bias_bits, d.bias)
        # ... and the two inputs of that rule from the residues themselves: rescore_isolated_domain(long_target = TRUE), the
        # envelope under its own length model with the profile's odds and with the odds of the composition-mixed background
        # (smoothing 25 / min(100, max(50, window length)): any of the oracle windows holding the envelope may be the one)
        env = seq[lo - 1:hi] if not rev else DNA_COMP[seq[lo - 1:hi][::-1]]
        ok = False
        nshortwin += min(wlens) < 100
        for wl in sorted(set(wlens)):
            orig, adj = oracle_lib.lt_envelope_scores(op, env, wl)
            if abs(orig - d.envelope_score * ln2) <= LT_ENV_TOL_NATS and abs(max(0.0, orig - adj) - d.correction * ln2) <= LT_ENV_TOL_NATS:
                ok = True
        assert ok, (h.name, d.env_from, d.env_to, orig, adj, d.envelope_score * ln2, d.correction * ln2, wlens)
    assert nshort >= min_short, nshort
    return (int(total[0]), nshort, nshortwin) if want_short_windows else (int(total[0]), nshort)


def check_hit_coordinates_against_oracle(pipeline, hmm, seq, hits, oracle=None, min_hits=1):
    """Every hit's envelope, alignment and model coordinates from the oracle's OWN long-target domain definition
    (oracle/p7_oracle_dd.c p7o_lt_domains) of a Forward-passing window that holds it: exact -- except that a window with a
    region resolved by sampled tracebacks may come out differently when a sampled choice falls on a tie of the two
    implementations' summation orders (every later sample of the region then differs): at most 5 % of the hits, all of them
    in such windows.  Returns (hits reproduced, of those in ensemble regions)."""
    op, max_length, units, total = oracle or oracle_windows(pipeline, hmm, seq)
    cache = {}
    checked = clustered = tolerated = 0
    for h in hits:
        d = h.domains[0]
        a = d.alignment
        # the strand column compares the alignment's ends and reads "-" when they are one residue (as upstream prints it):
        # the envelope's ends tell
        rev = d.env_from > d.env_to if d.env_from != d.env_to else d.strand == "-"
        lo, hi = min(d.env_from, d.env_to), max(d.env_from, d.env_to)
        want = (d.env_from, d.env_to, a.target_from, a.target_to, a.hmm_from, a.hmm_to)
        found = sampled = False
        for (i, n, strand), win in units.items():
            if strand != int(rev) or len(win) == 0 or found:
                continue
            b_lo, b_hi = (lo - i, hi - i) if not rev else (i + n - hi + 1, i + n - lo + 1)
            if b_lo < 1 or b_hi > n:
                continue
            blk = seq[i:i + n] if strand == 0 else DNA_COMP[seq[i:i + n][::-1]]
            for w in np.nonzero((win[:, 0] <= b_lo) & (win[:, 0] + win[:, 1] - 1 >= b_hi))[0]:
                ws, wl = int(win[w, 0]), int(win[w, 1])
                key = (i, n, strand, ws, wl)
                if key not in cache:
                    cache[key] = oracle_lib.lt_domains(op, blk[ws - 1:ws - 1 + wl], do_null2=pipeline.null2, seed=pipeline.seed)
                sampled = sampled or cache[key][1][2] > 0
                for e in cache[key][0]:
                    def pos(x):                                # window coordinate -> target coordinate on the hit's strand
                        b = ws - 1 + int(x)                    # block coordinate, 1-based
                        return i + b if not rev else i + n - b + 1
                    got = (pos(e[0]), pos(e[1]), pos(e[2]), pos(e[3]), int(e[4]), int(e[5]))
                    if got == want:
                        found = True
                        clustered += int(e[12] == 1)
        assert found or sampled, (h.name, want)
        checked += int(found)
        tolerated += int(not found)
    assert checked >= min_hits and tolerated <= max(2, len(hits) // 20), (checked, tolerated)
    return checked, clustered
