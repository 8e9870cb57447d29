"""Synthetic workloads of bench.py that need more than a few lines: the Pfam-shaped profile library and the
Swiss-Prot-shaped target database of SURVEY.md 8(d), configs 3 and 4.

Pfam-A is not available offline, so the library is derived from the 14 calibrated fixture protein models
(tests/golden/hmms: PF02826, Thioesterase, KR, LuxC, 10 x RREFam): entry e stretches or shrinks a template to a length
M ~ lognormal(median 120, sigma 0.8) clipped to [20, 2000] (the new model's nodes are the template's nodes
round(k * Mt / M), k = 1..M, in a per-entry random ORDER: entries that share a template keep its emission and
transition statistics but are not homologous to one another, as Pfam families are not), mixes a little Dirichlet
noise into every emission row, and is then calibrated the way hmmbuild
calibrates (p7_Calibrate: Gumbel location of the MSV / Viterbi scores and the exponential tail of the Forward scores
of 200 random 100-residue sequences), the scores coming from the device filters.  Targets: L ~ lognormal(mu 5.65,
sigma 0.65) clipped to [30, 5000], residues i.i.d. from the background, and every second target carries one domain
sampled from the match states of a random library entry, so that the later stages of the pipeline have work.
Bench infrastructure only: nothing here is imported by pyhmmer_amd.
"""
import math
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
TEMPLATES = ("PF02826", "Thioesterase", "KR", "LuxC", "RREFam")


def load_templates():
    from pyhmmer_amd import plan7
    out = []
    for name in TEMPLATES:
        with plan7.HMMFile(ROOT / "tests" / "golden" / "hmms" / f"{name}.hmm") as hf:
            out.extend(hf)
    return out


def library_lengths(n, seed=43):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.lognormal(math.log(120.0), 0.8, size=n)), 20, 2000).astype(np.int64)


def make_entry(templates, e, M, seed=43):
    """Library entry e: template e mod 14 resampled to M nodes, emissions perturbed."""
    from pyhmmer_amd import plan7
    tpl = templates[e % len(templates)]
    rng = np.random.default_rng([seed, e])
    K = tpl.alphabet.K
    src = np.clip(np.rint(np.arange(M + 1) * (tpl.M / M)), 1, tpl.M).astype(np.int64)
    src[1:] = src[1:][rng.permutation(M)]              # a family of its own: same node statistics, different order
    src[0] = 0
    hmm = plan7.HMM(tpl.alphabet, M, f"syn{e:05d}_{tpl.name}")
    t = tpl.transition_probabilities[src].astype(np.float64)
    t[M] = tpl.transition_probabilities[tpl.M]            # the last node keeps the end conventions (no M->D, D->D)
    mat = tpl.match_emissions[src].astype(np.float64)
    noise = rng.dirichlet(np.full(K, 0.5), size=M + 1)
    mat[1:] = 0.9 * mat[1:] + 0.1 * noise[1:]
    mat[1:] /= mat[1:].sum(axis=1, keepdims=True)
    hmm.transition_probabilities[:] = t
    hmm.match_emissions[:] = mat
    hmm.insert_emissions[:] = tpl.insert_emissions[src]
    hmm.composition = mat[1:].mean(axis=0).astype(np.float32)
    cons = tpl.consensus or ("x" * tpl.M)
    hmm.consensus = "".join(cons[s - 1] for s in src[1:])
    hmm._evparam[:] = tpl._evparam                        # placeholder until calibrate() ran
    hmm.max_length = int(tpl.max_length * M / tpl.M) if tpl.max_length else 4 * M
    return hmm


def _gumbel_fit_loc(x, lam):
    return -math.log(np.mean(np.exp(-lam * (x - x.min())))) / lam + x.min()


def _gumbel_fit_complete(x):
    """ML fit of (mu, lambda) (Easel esl_gumbel_FitComplete: Newton on lambda, then the location)."""
    x = np.asarray(x, dtype=np.float64)
    lam = math.pi / math.sqrt(6.0 * max(x.var(), 1e-6))
    for _ in range(100):
        w = np.exp(-lam * (x - x.min()))
        sw, swx, swxx = w.sum(), (w * x).sum(), (w * x * x).sum()
        f = 1.0 / lam - x.mean() + swx / sw
        df = -1.0 / (lam * lam) - (swxx / sw - (swx / sw) ** 2)
        step = f / df
        lam_new = lam - step
        if lam_new <= 0:
            lam_new = lam / 2
        if abs(lam_new - lam) < 1e-7:
            lam = lam_new
            break
        lam = lam_new
    return _gumbel_fit_loc(x, lam), lam


class Calibrator:
    """p7_Calibrate with the device filters as the scorer: 200 random sequences of 100 residues, resident once."""

    N, L = 200, 100

    def __init__(self, alphabet, device=0, seed=42):
        from pyhmmer_amd import plan7
        self.bg = plan7.Background(alphabet)
        f = self.bg.residue_frequencies.astype(np.float64)
        rng = np.random.default_rng(seed)
        res = rng.choice(alphabet.K, size=(self.N, self.L), p=f / f.sum()).astype(np.uint8)
        flat = np.concatenate([[255], np.concatenate([res, np.full((self.N, 1), 255, np.uint8)], axis=1).reshape(-1)]).astype(np.uint8)
        offsets = 1 + np.arange(self.N, dtype=np.int64) * (self.L + 1)
        self.db = plan7.SequenceDatabase.from_packed(alphabet, flat, offsets, np.full(self.N, self.L, np.int32), device=device)
        self.null1 = self.L * math.log(self.L / (self.L + 1.0)) + math.log(1.0 / (self.L + 1.0))
        self.bgf = f / f.sum()

    def calibrate(self, hmm):
        from pyhmmer_amd import plan7
        om = plan7.OptimizedProfile(hmm, self.bg, self.L)
        got = self.db.filters(om, msv=True, viterbi=True, forward=True)
        pm = math.log(3.0 / (self.L + 3.0))
        tjb = min(255.0, float(np.rint(-om.scale_b * pm)))
        usc = (got["xJ"].astype(np.float64) - tjb - om.base) / om.scale_b - 3.0
        vsc = (got["xC"].astype(np.float64) + float(np.rint(om.scale_w * pm)) - om.base_w) / om.scale_w - 3.0
        fsc = got["fwd"].astype(np.float64)
        bits = lambda s: (s - self.null1) / math.log(2.0)
        mat = np.maximum(hmm.match_emissions[1:].astype(np.float64), 1e-9)
        H = float(np.mean(np.sum(mat * np.log2(mat / self.bgf), axis=1)))        # p7_MeanMatchRelativeEntropy
        lam = math.log(2.0) + 1.44 / (hmm.M * max(H, 0.05))                     # p7_Lambda
        ok_m, ok_v = got["xJ"] >= 0, got["xC"] < 32767
        mmu = _gumbel_fit_loc(bits(usc[ok_m]), lam)
        vmu = _gumbel_fit_loc(bits(vsc[ok_v]), lam)
        gmu, glam = _gumbel_fit_complete(bits(fsc[np.isfinite(fsc)]))
        tailp = 0.04
        tau = gmu - math.log(-math.log(1.0 - tailp)) / glam + math.log(tailp) / lam     # p7_Tau
        hmm._evparam[:] = [mmu, lam, vmu, lam, tau, lam]
        return hmm


def make_library(n, device=0, seed=43, count=None):
    """The first `count` (default: all) calibrated entries of the n-entry library, and the lengths of all n."""
    templates = load_templates()
    lengths = library_lengths(n, seed)
    cal = Calibrator(templates[0].alphabet, device=device)
    count = n if count is None else min(count, n)
    return [cal.calibrate(make_entry(templates, e, int(lengths[e]), seed)) for e in range(count)], lengths, templates


def make_targets(nseq, library_size, templates, lib_lengths, seed=44, planted_frac=0.5, lib_seed=43):
    """Flat arrays in the C-ABI's input format (255 x1..xL 255 ...), lengths lognormal, a fraction `planted_frac` of the
    targets carrying one domain sampled from the match emissions of a random entry among the first `library_size`
    of the library.  (A Pfam-like hit density is ~12 targets per family: planted_frac = 12.5 * library_size / nseq;
    a run over the first few thousand entries plants only their domains -- the others' would be background to it.)"""
    from pyhmmer_amd import plan7
    rng = np.random.default_rng(seed)
    abc = templates[0].alphabet
    K = abc.K
    lens = np.clip(np.rint(rng.lognormal(5.65, 0.65, size=nseq)), 30, 5000).astype(np.int64)
    bg = plan7.Background(abc).residue_frequencies.astype(np.float64)
    cum = np.cumsum(bg / bg.sum())
    cum[-1] = 1.0
    lut = np.minimum(np.searchsorted(cum, (np.arange(65536) + 0.5) / 65536.0, side="right"), K - 1).astype(np.uint8)
    offsets = 1 + np.concatenate([[0], np.cumsum(lens[:-1] + 1)])
    flat = np.empty(int(offsets[-1] + lens[-1] + 1), dtype=np.uint8)
    flat[:] = lut[rng.integers(0, 65536, size=flat.shape[0], dtype=np.uint16)]
    flat[0] = 255
    flat[offsets + lens] = 255
    # planted domains, grouped by library entry so that one entry's emission table is built once
    planted = np.nonzero(rng.random(nseq) < planted_frac)[0]
    entry = rng.integers(0, library_size, size=planted.shape[0])
    order = np.argsort(entry, kind="stable")
    planted, entry = planted[order], entry[order]
    bounds = np.concatenate([[0], np.nonzero(np.diff(entry))[0] + 1, [entry.shape[0]]])
    nplanted = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        e = int(entry[a])
        hmm = make_entry(templates, e, int(lib_lengths[e]), lib_seed)
        cm = np.cumsum(hmm.match_emissions[1:].astype(np.float64), axis=1)
        cm /= cm[:, -1:]
        for tgt in planted[a:b]:
            L = int(lens[tgt])
            dom_len = min(hmm.M, L)
            k0 = int(rng.integers(0, hmm.M - dom_len + 1))
            u = rng.random(dom_len)
            dom = (u[:, None] > cm[k0:k0 + dom_len]).sum(axis=1).astype(np.uint8)
            start = int(rng.integers(0, L - dom_len + 1))
            o = int(offsets[tgt]) + start
            flat[o:o + dom_len] = np.minimum(dom, K - 1)
            nplanted += 1
    return flat, offsets.astype(np.int64), lens.astype(np.int32), nplanted


def make_chromosome(hmm, L, planted=50, seed=45, return_planted=False):
    """configs[4]: an i.i.d. ACGT (0.25 each) chromosome of L residues with `planted` mutated (20 %) stretches of the
    model's consensus, every second one on the reverse strand, so that the stages behind the SSV scan have work.
    return_planted: also the stretches as (position, length, strand, first model node), 0-based."""
    rng = np.random.default_rng(seed)
    where = []
    seq = rng.integers(0, 4, size=L, dtype=np.uint8)
    cons = np.argmax(hmm.match_emissions[1:], axis=1).astype(np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    for c in range(planted):
        a = int(rng.integers(0, max(1, hmm.M - 200)))
        n = min(int(rng.integers(150, 1200)), hmm.M - a)
        pos = int(rng.integers(0, L - n))
        seg = cons[a:a + n].copy()
        mut = rng.random(n) < 0.2
        seg[mut] = rng.integers(0, 4, size=int(mut.sum()))
        seq[pos:pos + n] = seg if c % 2 == 0 else comp[seg[::-1]]
        where.append((pos, n, c % 2, a))
    return (seq, where) if return_planted else seq
