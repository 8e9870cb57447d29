"""ctypes binding of the CPU oracle (oracle/_build/libp7oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by pyhmmer_amd."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
ROOT = ORACLE_DIR.parent
LIB = ORACLE_DIR / "_build" / "libp7oracle.so"


class Profile(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("K", C.c_int), ("Kp", C.c_int), ("Q16", C.c_int), ("Q8", C.c_int), ("Q4", C.c_int),
        ("L", C.c_int), ("nj", C.c_float),
        ("tsc", C.POINTER(C.c_float)), ("msc", C.POINTER(C.c_float)), ("xsc", (C.c_float * 2) * 4),
        ("rbv", C.POINTER(C.c_uint8)), ("sbv", C.POINTER(C.c_int8)),
        ("tbm_b", C.c_uint8), ("tec_b", C.c_uint8), ("tjb_b", C.c_uint8), ("base_b", C.c_uint8), ("bias_b", C.c_uint8),
        ("scale_b", C.c_float),
        ("rwv", C.POINTER(C.c_int16)), ("twv", C.POINTER(C.c_int16)), ("xw", (C.c_int16 * 2) * 4),
        ("scale_w", C.c_float), ("base_w", C.c_int16), ("ddbound_w", C.c_int16), ("ncj_roundoff", C.c_float),
        ("rfv", C.POINTER(C.c_float)), ("tfv", C.POINTER(C.c_float)), ("xf", (C.c_float * 2) * 4),
        ("evparam", C.c_float * 6), ("compo", C.c_float * 20), ("bgf", C.c_float * 20),
    ]


class Record(C.Structure):
    _fields_ = [
        ("usc", C.c_float), ("filtersc", C.c_float), ("nullsc", C.c_float), ("vfsc", C.c_float), ("fwdsc", C.c_float),
        ("P_msv", C.c_double), ("P_bias", C.c_double), ("P_vit", C.c_double), ("P_fwd", C.c_double),
        ("xJ_msv", C.c_int32), ("xC_vit", C.c_int32), ("stage", C.c_int32), ("ran_vit", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("nseqs", "nres", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd")]


_lib = None


def build():
    subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        l = C.CDLL(str(LIB))
        PP = C.POINTER(Profile)
        l.p7o_profile_build.restype = PP
        l.p7o_profile_build.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        l.p7o_profile_free.argtypes = [PP]
        l.p7o_reconfig_length.argtypes = [PP, C.c_int]
        for fn in ("p7o_msv", "p7o_msv_scalar", "p7o_vit", "p7o_vit_scalar"):
            getattr(l, fn).argtypes = [PP, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        l.p7o_fwd.argtypes = [PP, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
        l.p7o_bck.argtypes = [PP, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        l.p7o_null1.restype = C.c_float
        l.p7o_null1.argtypes = [C.c_int]
        l.p7o_bias_filter.restype = C.c_float
        l.p7o_bias_filter.argtypes = [PP, C.c_void_p, C.c_int]
        l.p7o_cascade.argtypes = [PP, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(Record)]
        l.p7o_cascade_block.argtypes = [PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double,
                                        C.c_double, C.c_int, C.c_void_p, C.POINTER(Counters)]
        l.p7o_msv_block.argtypes = [PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.p7o_expf_neg.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.p7o_sse_expf_scalar.restype = C.c_float
        l.p7o_sse_expf_scalar.argtypes = [C.c_float]
        for fn in ("p7o_gumbel_surv", "p7o_exp_surv", "p7o_exp_logsurv"):
            getattr(l, fn).restype = C.c_double
            getattr(l, fn).argtypes = [C.c_double] * 3
        _lib = l
    return _lib


class OracleProfile:
    """Owns a P7O_PROFILE built from a pyhmmer_amd.plan7.HMM (host-side parser) + background."""

    def __init__(self, hmm, bg, L=400):
        l = lib()
        t = np.ascontiguousarray(hmm.transition_probabilities, dtype=np.float32)
        mat = np.ascontiguousarray(hmm.match_emissions, dtype=np.float32)
        bgf = np.ascontiguousarray(bg.residue_frequencies, dtype=np.float32)
        compo = None if hmm.composition is None else np.ascontiguousarray(hmm.composition, dtype=np.float32)
        ev = np.ascontiguousarray(hmm._evparam, dtype=np.float32)
        self.ptr = l.p7o_profile_build(hmm.M, hmm.alphabet.K, t.ctypes.data, mat.ctypes.data, bgf.ctypes.data,
                                       None if compo is None else compo.ctypes.data, ev.ctypes.data, L)
        self.p = self.ptr.contents
        self.hmm = hmm

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().p7o_profile_free(self.ptr)
            self.ptr = None

    def arr(self, name):
        p = self.p
        shapes = {
            "rbv": (p.Kp, p.Q16 * 16), "sbv": (p.Kp, (p.Q16 + 17) * 16), "rwv": (p.Kp, p.Q8 * 8),
            "twv": (8 * p.Q8, 8), "rfv": (p.Kp, p.Q4 * 4), "tfv": (8 * p.Q4, 4),
        }
        return np.ctypeslib.as_array(getattr(p, name), shape=shapes[name]).copy()

    @staticmethod
    def _dsq(seq):
        a = np.empty(len(seq) + 2, dtype=np.uint8)
        a[0] = a[-1] = 255
        a[1:-1] = seq
        return a

    def msv(self, seq, scalar=False):
        d = self._dsq(seq); sc = C.c_float(); xj = C.c_int()
        lib().p7o_reconfig_length(self.ptr, len(seq))
        st = (lib().p7o_msv_scalar if scalar else lib().p7o_msv)(self.ptr, d.ctypes.data, len(seq), C.byref(sc), C.byref(xj))
        return st, sc.value, xj.value

    def vit(self, seq, scalar=False):
        d = self._dsq(seq); sc = C.c_float(); xc = C.c_int()
        lib().p7o_reconfig_length(self.ptr, len(seq))
        st = (lib().p7o_vit_scalar if scalar else lib().p7o_vit)(self.ptr, d.ctypes.data, len(seq), C.byref(sc), C.byref(xc))
        return st, sc.value, xc.value

    def fwd(self, seq, want_xmx=False):
        d = self._dsq(seq); sc = C.c_float()
        lib().p7o_reconfig_length(self.ptr, len(seq))
        xmx = np.zeros((len(seq) + 1, 6), dtype=np.float32) if want_xmx else None
        st = lib().p7o_fwd(self.ptr, d.ctypes.data, len(seq), None if xmx is None else xmx.ctypes.data, C.byref(sc))
        return (st, sc.value, xmx) if want_xmx else (st, sc.value)

    def bck(self, seq):
        st, fsc, fx = self.fwd(seq, want_xmx=True)
        d = self._dsq(seq); sc = C.c_float()
        bx = np.zeros_like(fx)
        st = lib().p7o_bck(self.ptr, d.ctypes.data, len(seq), fx.ctypes.data, bx.ctypes.data, C.byref(sc))
        return st, sc.value, fx, bx

    def bias(self, seq):
        d = self._dsq(seq)
        return lib().p7o_bias_filter(self.ptr, d.ctypes.data, len(seq))

    def cascade_block(self, packed, F1=0.02, F2=1e-3, F3=1e-5, do_bias=True, want_records=True):
        n = packed.n
        recs = (Record * n)() if want_records else None
        ctr = Counters()
        lib().p7o_cascade_block(self.ptr, packed.dsq.ctypes.data, packed.offsets.ctypes.data, packed.lengths.ctypes.data,
                                n, F1, F2, F3, int(do_bias), recs, C.byref(ctr))
        return recs, ctr

    def msv_block(self, packed):
        out = np.empty(packed.n, dtype=np.int32)
        lib().p7o_msv_block(self.ptr, packed.dsq.ctypes.data, packed.offsets.ctypes.data, packed.lengths.ctypes.data,
                            packed.n, out.ctypes.data)
        return out


def ssv_longtarget(op, block_dsq, max_length, F1=0.02, cap=1 << 16):
    """Oracle's exact sequential p7_SSVFilter_longtarget over one strand block (block_dsq: residues, 0-based numpy)."""
    l = lib()
    l.p7o_ssv_longtarget.restype = C.c_int64
    l.p7o_ssv_longtarget.argtypes = [C.POINTER(Profile), C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_int64]
    d = np.empty(len(block_dsq) + 2, dtype=np.uint8)
    d[0] = d[-1] = 255
    d[1:-1] = block_dsq
    seeds = np.zeros((cap, 3), dtype=np.int64)
    n = l.p7o_ssv_longtarget(op.ptr, d.ctypes.data, len(block_dsq), int(max_length), float(F1), seeds.ctypes.data, cap)
    assert n <= cap
    return seeds[:n].copy()


def lt_block(op, block_dsq, max_length, F1=0.02, F2=3e-3, F3=3e-5, B1=110, B2=240, B3=1000, do_bias=True, cap=1 << 14):
    """The tail of p7_Pipeline_LongTarget for one strand block (p7_oracle_lt.c).  Returns (windows[n, 5], counts[8]): the windows
    that pass Forward as (first residue, length, fwdsc, nullsc, filtersc at F3) in block coordinates; counts = windows past
    MSV, bias, Viterbi, Forward, then residues past each."""
    l = lib()
    l.p7o_lt_block.restype = C.c_int64
    l.p7o_lt_block.argtypes = [C.POINTER(Profile), C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    d = np.empty(len(block_dsq) + 2, dtype=np.uint8)
    d[0] = d[-1] = 255
    d[1:-1] = block_dsq
    out = np.zeros((cap, 6), dtype=np.float64)
    counts = np.zeros(8, dtype=np.uint64)
    n = l.p7o_lt_block(op.ptr, d.ctypes.data, len(block_dsq), int(max_length), F1, F2, F3, B1, B2, B3, int(do_bias),
                       out.ctypes.data, cap, counts.ctypes.data)
    assert n <= cap
    return out[:n, :5].copy(), counts


def lt_domain_score(op, max_length, env_len, ali_len, envsc, domcorrection, do_null2=True):
    l = lib()
    l.p7o_lt_domain_score.restype = C.c_float
    l.p7o_lt_domain_score.argtypes = [C.POINTER(Profile), C.c_int, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    bias, lnp = C.c_float(0), C.c_double(0)
    sc = l.p7o_lt_domain_score(op.ptr, int(max_length), int(env_len), int(ali_len), float(envsc), float(domcorrection), int(do_null2),
                               C.byref(bias), C.byref(lnp))
    return float(sc), float(bias.value), float(lnp.value)


def lt_envelope_scores(op, env_residues, window_len):
    """Forward of one envelope under its own length model, unihit: (orig, adjusted) in nats -- with the profile's odds and with
    odds re-derived for the background mixed with the envelope's composition (reparameterize_model)."""
    l = lib()
    l.p7o_lt_envelope_scores.restype = C.c_int
    l.p7o_lt_envelope_scores.argtypes = [C.POINTER(Profile), C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    d = np.empty(len(env_residues) + 2, dtype=np.uint8)
    d[0] = d[-1] = 255
    d[1:-1] = env_residues
    K = op.ptr.contents.K
    degen = np.zeros((32, K), dtype=np.uint8)
    degen[:K, :K] = np.eye(K, dtype=np.uint8)
    orig, adj = C.c_float(0), C.c_float(0)
    st = l.p7o_lt_envelope_scores(op.ptr, d.ctypes.data, len(env_residues), int(window_len), degen.ctypes.data, C.byref(orig), C.byref(adj))
    assert st == 0
    return float(orig.value), float(adj.value)


# Easel's standard residue-code sets (esl_alphabet.c: B = ND, J = IL, Z = QE, O = K, U = C, X = any; R = AG, Y = CT, ...)
_DEGEN_AMINO = {21: "ND", 22: "IL", 23: "QE", 24: "K", 25: "C", 26: "ACDEFGHIKLMNPQRSTVWY"}
_DEGEN_NUCLEIC = {5: "AG", 6: "CT", 7: "AC", 8: "GT", 9: "CG", 10: "AT", 11: "ACT", 12: "CGT", 13: "ACG", 14: "AGT", 15: "ACGT"}


def degeneracy_sets(K):
    """[Kp][K] matrix: which canonical residues a residue code stands for."""
    Kp = 29 if K == 20 else 18
    canon = "ACDEFGHIKLMNPQRSTVWY" if K == 20 else "ACGT"
    d = np.zeros((Kp, K), dtype=np.uint8)
    d[:K, :K] = np.eye(K, dtype=np.uint8)
    for x, members in (_DEGEN_AMINO if K == 20 else _DEGEN_NUCLEIC).items():
        for c in members:
            d[x, canon.index(c)] = 1
    return d


def domains(op, seq, do_null2=True, seed=42, ensembles=True, want_sequence=False):
    """p7_oracle_dd.c on one target: (envelopes, counts).  envelopes: rows of ienv jenv iali jali hmmfrom hmmto envsc(nats)
    domcorrection(nats) oasc bitscore(bits) dombias(bits) lnP kind (0: a region that holds one domain, 1: a cluster of an
    ensemble region), in the reference's order; counts = (regions, envelopes, ensemble regions, clusters, overlapping
    clusters) -- the reference's (nregions, nclustered, noverlaps, nenvelopes) are counts[0], [2], [4], [1].
    ensembles=False: regions that need the traceback ensemble are counted and left out.  want_sequence: also the
    sequence's scores as p7_pipeline.c derives them (bit score, pre-score, sum-of-domains score, ln P, domains, their length)."""
    l = lib()
    l.p7o_domains.restype = C.c_int64
    l.p7o_domains.argtypes = [C.POINTER(Profile), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int,
                              C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p]
    st, bsc, fx, bx = op.bck(seq)                          # configures the profile for len(seq)
    fst, fwdsc = op.fwd(seq)
    d = op._dsq(seq)
    degen = degeneracy_sets(op.ptr.contents.K)
    cap = 256
    out = np.zeros((cap, 13), dtype=np.float64)
    counts = np.zeros(5, dtype=np.int64)
    seqout = np.zeros(6, dtype=np.float64)
    n = l.p7o_domains(op.ptr, d.ctypes.data, len(seq), fx.ctypes.data, bx.ctypes.data, degen.ctypes.data, int(do_null2), int(seed),
                      int(bool(ensembles)), out.ctypes.data, cap, counts.ctypes.data, float(fwdsc), seqout.ctypes.data)
    assert 0 <= n <= cap, n
    if want_sequence:
        return out[:n].copy(), tuple(int(c) for c in counts), dict(zip(("score", "pre_score", "sum_score", "lnP", "ndom", "domain_residues"), seqout.tolist()))
    return out[:n].copy(), tuple(int(c) for c in counts)


def domains_single(op, seq, do_null2=True):
    """The regions that hold one domain only: (envelopes, (regions, single-domain envelopes, ensemble regions))."""
    envs, counts = domains(op, seq, do_null2=do_null2, ensembles=False)
    return envs, (counts[0], counts[1], counts[2])


def domain_alignment(op, seq, ienv, jenv):
    """p7_alidisplay_Create of the envelope's optimal-accuracy alignment: (model, match, sequence, posterior) lines."""
    l = lib()
    l.p7o_domain_alignment.restype = C.c_int
    l.p7o_domain_alignment.argtypes = [C.POINTER(Profile), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p,
                                       C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    l.p7o_reconfig_length(op.ptr, len(seq))
    d = op._dsq(seq)
    hmm = op.hmm
    cap = 2 * (len(seq) + hmm.M) + 16
    bufs = [C.create_string_buffer(cap) for _ in range(4)]
    n = l.p7o_domain_alignment(op.ptr, d.ctypes.data, len(seq), int(ienv), int(jenv), (" " + hmm.consensus).encode(),
                               hmm.alphabet.symbols.encode(), bufs[0], bufs[1], bufs[2], bufs[3], cap)
    assert 0 <= n < cap, n
    return tuple(b.value.decode() for b in bufs)


def lt_domains(op, window, do_null2=True, seed=42, max_env_extra=20):
    """Domain definition of one Forward-passing window of a long target (p7o_lt_domains): rows of ienv jenv iali jali hmmfrom
    hmmto envsc domcorrection oasc ... kind in window coordinates, and the counters."""
    l = lib()
    l.p7o_lt_domains.restype = C.c_int64
    l.p7o_lt_domains.argtypes = [C.POINTER(Profile), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int,
                                 C.c_void_p, C.c_int64, C.c_void_p]
    st, bsc, fx, bx = op.bck(window)                       # configures the profile for the window's length
    d = op._dsq(window)
    degen = degeneracy_sets(op.ptr.contents.K)
    cap = 256
    out = np.zeros((cap, 13), dtype=np.float64)
    counts = np.zeros(5, dtype=np.int64)
    n = l.p7o_lt_domains(op.ptr, d.ctypes.data, len(window), fx.ctypes.data, bx.ctypes.data, degen.ctypes.data, int(do_null2), int(seed),
                         int(max_env_extra), out.ctypes.data, cap, counts.ctypes.data)
    assert 0 <= n <= cap, n
    return out[:n].copy(), tuple(int(c) for c in counts)
