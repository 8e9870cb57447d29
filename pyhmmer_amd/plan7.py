"""Host-side mirror of the ``pyhmmer.plan7`` classes that sit on the ``p7_Pipeline`` hot path.

Same names, argument meanings and error behaviour as the reference (``src/pyhmmer/plan7.pyx``):
``HMM`` / ``HMMFile`` (text HMMER3 reader only), ``Background``, ``Profile``, ``OptimizedProfile``,
``Pipeline`` (``search_hmm``), ``TopHits`` / ``Hit`` / ``Domain`` / ``Alignment``.  All compute goes
through the C-ABI of ``libp7x.so`` (``include/p7x.h``); there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Iterable, Iterator, List, Optional, Union

import numpy as np

from . import _lib
from .easel import (Alphabet, DigitalSequence, DigitalSequenceBlock, SequenceFile, eslAMINO, eslDNA,
                    eslRNA)
from .errors import (AlphabetMismatch, AllocationError, InvalidParameter, MissingCutoffs,
                     UnexpectedError, status_to_exception)

__all__ = [
    "HMM", "HMMFile", "Background", "Profile", "OptimizedProfile", "EvalueParameters", "Cutoffs",
    "Pipeline", "TopHits", "Hit", "Domain", "Domains", "Alignment",
]

CUTOFF_UNSET = -99999.0
EVPARAM_UNSET = -99999.0

# p7_AminoFrequencies (upstream p7_bg.c; cross-checked against the insert emissions of
# tests/golden/hmms/Thioesterase.hmm) -- SURVEY.md section 8 row a5
_AMINO_BG = np.array([
    0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198, 0.0590092,
    0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639, 0.0540978, 0.0683364,
    0.0540687, 0.0673417, 0.0114135, 0.0304133], dtype=np.float32)


def _fptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


# --------------------------------------------------------------------------- small views

class EvalueParameters:
    """Reference ``plan7.pyx:1689-1849``."""
    __slots__ = ("_v",)
    _names = ("m_mu", "m_lambda", "v_mu", "v_lambda", "f_tau", "f_lambda")

    def __init__(self, values):
        self._v = np.asarray(values, dtype=np.float32)

    def __getattr__(self, name):
        try:
            v = float(self._v[self._names.index(name)])
        except ValueError:
            raise AttributeError(name)
        return None if v == EVPARAM_UNSET else v

    def as_vector(self) -> np.ndarray:
        return self._v.copy()

    def __repr__(self):
        return "<EvalueParameters " + " ".join(f"{n}={getattr(self, n)!r}" for n in self._names) + ">"


class Cutoffs:
    """Reference ``plan7.pyx:1154-1440``."""
    __slots__ = ("_v",)

    def __init__(self, values):
        self._v = np.asarray(values, dtype=np.float32)

    def _pair(self, i):
        a, b = float(self._v[i]), float(self._v[i + 1])
        return None if a == CUTOFF_UNSET or b == CUTOFF_UNSET else (a, b)

    @property
    def gathering(self):
        return self._pair(0)

    @property
    def trusted(self):
        return self._pair(2)

    @property
    def noise(self):
        return self._pair(4)

    def gathering_available(self) -> bool:
        return self.gathering is not None

    def trusted_available(self) -> bool:
        return self.trusted is not None

    def noise_available(self) -> bool:
        return self.noise is not None

    def as_vector(self) -> np.ndarray:
        return self._v.copy()


# --------------------------------------------------------------------------- Background

class Background:
    """The null model (reference ``plan7.pyx:427-603``; ``P7_BG`` in ``p7_bg.pxd:10-30``)."""

    def __init__(self, alphabet: Alphabet, uniform: bool = False):
        self.alphabet = alphabet
        self.uniform = uniform
        if uniform or not alphabet.is_amino():
            self.residue_frequencies = np.full(alphabet.K, 1.0 / alphabet.K, dtype=np.float32)
        else:
            self.residue_frequencies = _AMINO_BG.copy()
        self._L = 350
        self.omega = 1.0 / 256.0

    @property
    def L(self) -> int:
        return self._L

    @L.setter
    def L(self, L: int):
        self._L = int(L)

    @property
    def transition_probability(self) -> float:
        return float(np.float32(self._L) / np.float32(self._L + 1))

    def copy(self) -> "Background":
        b = Background(self.alphabet, self.uniform)
        b.residue_frequencies = self.residue_frequencies.copy()
        b._L = self._L
        return b


# --------------------------------------------------------------------------- HMM

class HMM:
    """A core profile HMM (reference ``plan7.pyx:2236-3655``; ``P7_HMM`` in ``p7_hmm.pxd:48-78``).

    ``transition_probabilities`` is ``(M+1, 7)`` in the order MM, MI, MD, IM, II, DM, DD;
    ``match_emissions`` / ``insert_emissions`` are ``(M+1, K)``.
    """

    def __init__(self, alphabet: Alphabet, M: int, name: str):
        self.alphabet = alphabet
        self.M = int(M)
        self.name = name
        self.accession: Optional[str] = None
        self.description: Optional[str] = None
        K = alphabet.K
        self.transition_probabilities = np.zeros((M + 1, 7), dtype=np.float32)
        self.match_emissions = np.zeros((M + 1, K), dtype=np.float32)
        self.insert_emissions = np.zeros((M + 1, K), dtype=np.float32)
        self.composition: Optional[np.ndarray] = None
        self.consensus: Optional[str] = None
        self.consensus_structure: Optional[str] = None
        self.reference: Optional[str] = None
        self.model_mask: Optional[str] = None
        self.map: Optional[np.ndarray] = None
        self._evparam = np.full(6, EVPARAM_UNSET, dtype=np.float32)
        self._cutoffs = np.full(6, CUTOFF_UNSET, dtype=np.float32)
        self.nseq: Optional[int] = None
        self.nseq_effective: Optional[float] = None
        self.max_length: Optional[int] = None
        self.checksum: Optional[int] = None
        self.command_line: Optional[str] = None
        self.creation_time: Optional[str] = None

    @property
    def evalue_parameters(self) -> EvalueParameters:
        return EvalueParameters(self._evparam)

    @property
    def cutoffs(self) -> Cutoffs:
        return Cutoffs(self._cutoffs)

    def __repr__(self):
        return f"<HMM name={self.name!r} M={self.M} alphabet={self.alphabet!r}>"

    def _view(self):
        """Build the ``p7x_hmm_view`` (keeps the numpy buffers alive on the returned tuple)."""
        t = np.ascontiguousarray(self.transition_probabilities, dtype=np.float32)
        mat = np.ascontiguousarray(self.match_emissions, dtype=np.float32)
        ins = np.ascontiguousarray(self.insert_emissions, dtype=np.float32)
        compo = None if self.composition is None else np.ascontiguousarray(self.composition, dtype=np.float32)
        v = _lib.HmmView()
        v.M = self.M
        v.abc_type = self.alphabet.type_code
        v.t, v.mat, v.ins, v.compo = _fptr(t), _fptr(mat), _fptr(ins), _fptr(compo)
        for i in range(6):
            v.evparam[i] = float(self._evparam[i])
            v.cutoff[i] = float(self._cutoffs[i])
        v.max_length = -1 if self.max_length is None else int(self.max_length)
        v.name = self.name.encode()
        v.acc = None if self.accession is None else self.accession.encode()
        v.desc = None if self.description is None else self.description.encode()

        def ann(s):
            return None if s is None else (" " + s + "\0").encode()[: self.M + 2]
        v.consensus = ann(self.consensus)
        v.rf, v.mm, v.cs = ann(self.reference), ann(self.model_mask), ann(self.consensus_structure)
        return v, (t, mat, ins, compo)


class HMMFile:
    """Reader for HMMER3 ASCII save files (reference ``plan7.pyx:3656-4050``; upstream
    ``p7_hmmfile.c:read_asc30hmm``).  Binary ``.h3m`` and pressed databases are not read here;
    the pressed ``.h3f/.h3p`` fixtures are parsed only by the tests (``tests/h3_reader.py``)."""

    def __init__(self, file, db: bool = False, *, alphabet: Optional[Alphabet] = None):
        if isinstance(file, (str, bytes, os.PathLike)):
            self._fh = open(file, "r")
            self._own = True
            self.name = os.fspath(file)
        else:
            self._fh = file
            self._own = False
            self.name = getattr(file, "name", None)
        self._alphabet = alphabet

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._own:
            self._fh.close()

    def __iter__(self) -> Iterator[HMM]:
        return self

    def __next__(self) -> HMM:
        hmm = self.read()
        if hmm is None:
            raise StopIteration
        return hmm

    def rewind(self):
        self._fh.seek(0)

    @staticmethod
    def _probs(tokens: List[str]) -> np.ndarray:
        vals = np.array([math.inf if t == "*" else float(t) for t in tokens], dtype=np.float64)
        out = np.empty(vals.shape[0], dtype=np.float32)
        _lib.lib().p7x_expf_neg(vals.ctypes.data, out.ctypes.data, vals.shape[0])
        return out

    def read(self) -> Optional[HMM]:
        fh = self._fh
        line = fh.readline()
        while line and not line.strip():
            line = fh.readline()
        if not line:
            return None
        if not line.startswith("HMMER3/"):
            raise ValueError(f"Invalid format in file: {self.name!r} (expected an HMMER3 ASCII header)")
        fmt = line.split()[0]
        n_ann = {"HMMER3/f": 5, "HMMER3/e": 4, "HMMER3/d": 3, "HMMER3/c": 3, "HMMER3/b": 3, "HMMER3/a": 3}.get(fmt, 5)
        hdr = {}
        stats = {}
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("premature end of HMM file")
            tag = line[:5].strip()
            if tag == "HMM":
                break
            val = line[5:].strip()
            if tag == "STATS":
                f = val.split()
                stats[f[1]] = (float(f[2]), float(f[3]))
            else:
                hdr[tag] = val
        abc_name = hdr.get("ALPH", "amino").lower()
        alphabet = {"amino": Alphabet.amino(), "dna": Alphabet.dna(), "rna": Alphabet.rna()}[abc_name]
        if self._alphabet is not None and self._alphabet != alphabet:
            raise AlphabetMismatch(self._alphabet, alphabet)
        M = int(hdr["LENG"])
        K = alphabet.K
        hmm = HMM(alphabet, M, hdr["NAME"])
        hmm.accession = hdr.get("ACC")
        hmm.description = hdr.get("DESC")
        if "MAXL" in hdr:
            hmm.max_length = int(hdr["MAXL"])
        if "NSEQ" in hdr:
            hmm.nseq = int(hdr["NSEQ"])
        if "EFFN" in hdr:
            hmm.nseq_effective = float(hdr["EFFN"])
        if "CKSUM" in hdr:
            hmm.checksum = int(hdr["CKSUM"])
        hmm.command_line = hdr.get("COM")
        hmm.creation_time = hdr.get("DATE")
        for key, idx in (("GA", 0), ("TC", 2), ("NC", 4)):
            if key in hdr:
                a, b = hdr[key].rstrip(";").split()[:2]
                hmm._cutoffs[idx], hmm._cutoffs[idx + 1] = float(a), float(b.rstrip(";"))
        for key, idx in (("MSV", 0), ("VITERBI", 2), ("FORWARD", 4)):
            if key in stats:
                hmm._evparam[idx], hmm._evparam[idx + 1] = stats[key]
        flags = {k: hdr.get(k, "no").lower() == "yes" for k in ("RF", "MM", "CONS", "CS", "MAP")}
        fh.readline()                      # the "m->m m->i ..." column header
        line = fh.readline()
        tok = line.split()
        if tok[0] == "COMPO":
            hmm.composition = self._probs(tok[1:1 + K])
            line = fh.readline()
            tok = line.split()
        hmm.insert_emissions[0] = self._probs(tok[:K])
        hmm.transition_probabilities[0] = self._probs(fh.readline().split()[:7])
        hmm.match_emissions[0, 0] = 1.0    # upstream convention: mat[0] = {1,0,0,...}
        cons, rf, mm, cs = [], [], [], []
        mapv = np.zeros(M + 1, dtype=np.int64)
        for k in range(1, M + 1):
            tok = fh.readline().split()
            if int(tok[0]) != k:
                raise ValueError(f"expected match line for node {k}, found {tok[0]!r}")
            hmm.match_emissions[k] = self._probs(tok[1:1 + K])
            ann = tok[1 + K:]
            if flags["MAP"] and ann and ann[0] != "-":
                mapv[k] = int(ann[0])
            if n_ann >= 5:
                cons.append(ann[1]); rf.append(ann[2]); mm.append(ann[3]); cs.append(ann[4])
            elif n_ann == 4:
                cons.append(ann[1]); rf.append(ann[2]); cs.append(ann[3])
            else:
                rf.append(ann[1]); cs.append(ann[2])
            hmm.insert_emissions[k] = self._probs(fh.readline().split()[:K])
            hmm.transition_probabilities[k] = self._probs(fh.readline().split()[:7])
        line = fh.readline()
        if not line.startswith("//"):
            raise ValueError("expected // at end of HMM")
        if flags["CONS"] and cons:
            hmm.consensus = "".join(cons)
        if flags["RF"] and rf:
            hmm.reference = "".join(rf)
        if flags["MM"] and mm:
            hmm.model_mask = "".join(mm)
        if flags["CS"] and cs:
            hmm.consensus_structure = "".join(cs)
        if flags["MAP"]:
            hmm.map = mapv
        if hmm.consensus is None:
            hmm.consensus = _set_consensus(hmm)
        return hmm


def _set_consensus(hmm: HMM) -> str:
    """upstream p7_hmm_SetConsensus (no CONS annotation in the file): argmax emission, upper-case
    when its probability reaches 0.5 (amino) / 0.9 (nucleic)."""
    thresh = 0.5 if hmm.alphabet.is_amino() else 0.9
    out = []
    for k in range(1, hmm.M + 1):
        x = int(np.argmax(hmm.match_emissions[k]))
        c = hmm.alphabet.symbols[x]
        out.append(c.upper() if hmm.match_emissions[k, x] >= thresh else c.lower())
    return "".join(out)


# --------------------------------------------------------------------------- Profile / OptimizedProfile

class OptimizedProfile:
    """The search-ready query (reference ``plan7.pyx:4392-5070``).

    Owns a ``p7x_oprofile`` handle; the striped ``rbv/sbv/rwv/twv/rfv/tfv`` views the reference
    exposes (``plan7.pyx:4623-4813``) are re-created on demand from the un-striped device-oriented
    storage.
    """

    def __init__(self, hmm: HMM, background: Background, L: int = 400):
        self.alphabet = hmm.alphabet
        self._handle = C.c_void_p()
        view, keep = hmm._view()
        bgf = np.ascontiguousarray(background.residue_frequencies, dtype=np.float32)
        st = _lib.lib().p7x_oprofile_create(C.byref(view), bgf.ctypes.data, int(L), C.byref(self._handle))
        if st != 0:
            raise status_to_exception(st, "p7x_oprofile_create", _lib.last_error())
        self._info = _lib.OprofileInfo()
        _lib.lib().p7x_oprofile_get_info(self._handle, C.byref(self._info))
        self.name, self.accession, self.description = hmm.name, hmm.accession, hmm.description
        self.consensus = hmm.consensus
        self._hmm = hmm

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.lib().p7x_oprofile_destroy(h)
            except Exception:
                pass
            self._handle = None

    # scalars (reference plan7.pyx:4450-4620, 4726-4764)
    @property
    def M(self) -> int:
        return self._info.M

    @property
    def L(self) -> int:
        return self._info.L

    @property
    def tbm(self) -> int:
        return self._info.tbm_b

    @property
    def tec(self) -> int:
        return self._info.tec_b

    @property
    def tjb(self) -> int:
        return self._info.tjb_b

    @property
    def base(self) -> int:
        return self._info.base_b

    @property
    def bias(self) -> int:
        return self._info.bias_b

    @property
    def scale_b(self) -> float:
        return self._info.scale_b

    @property
    def scale_w(self) -> float:
        return self._info.scale_w

    @property
    def base_w(self) -> int:
        return self._info.base_w

    @property
    def ddbound_w(self) -> int:
        return self._info.ddbound_w

    @property
    def xw(self) -> np.ndarray:
        return np.array([[self._info.xw[i][j] for j in range(2)] for i in range(4)], dtype=np.int16)

    @property
    def xf(self) -> np.ndarray:
        return np.array([[self._info.xf[i][j] for j in range(2)] for i in range(4)], dtype=np.float32)

    @property
    def evalue_parameters(self) -> EvalueParameters:
        return EvalueParameters([self._info.evparam[i] for i in range(6)])

    @property
    def cutoffs(self) -> Cutoffs:
        return Cutoffs([self._info.cutoff[i] for i in range(6)])

    @property
    def compositions(self) -> np.ndarray:
        return np.array([self._info.compo[i] for i in range(self.alphabet.K)], dtype=np.float32)

    def _striped(self, which: int, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        n = _lib.lib().p7x_oprofile_striped(self._handle, which, out.ctypes.data, out.nbytes)
        if n != out.nbytes:
            raise UnexpectedError(int(n), "p7x_oprofile_striped")
        return out

    @property
    def rbv(self) -> np.ndarray:
        return self._striped(0, np.uint8, (self.alphabet.Kp, self._info.Q16 * 16))

    @property
    def sbv(self) -> np.ndarray:
        return self._striped(1, np.int8, (self.alphabet.Kp, (self._info.Q16 + 17) * 16))

    @property
    def rwv(self) -> np.ndarray:
        return self._striped(2, np.int16, (self.alphabet.Kp, self._info.Q8 * 8))

    @property
    def twv(self) -> np.ndarray:
        return self._striped(3, np.int16, (8 * self._info.Q8, 8))

    @property
    def rfv(self) -> np.ndarray:
        return self._striped(4, np.float32, (self.alphabet.Kp, self._info.Q4 * 4))

    @property
    def tfv(self) -> np.ndarray:
        return self._striped(5, np.float32, (8 * self._info.Q4, 4))

    # single-sequence filters (reference plan7.pyx:4969-5070)
    def _one(self, fn, seq: DigitalSequence, device: int = 0) -> float:
        if seq.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, seq.alphabet)
        sc = C.c_float()
        dsq = np.ascontiguousarray(seq.sequence, dtype=np.uint8)
        st = fn(self._handle, device, dsq.ctypes.data, dsq.shape[0], C.byref(sc))
        if st == 16:                       # eslERANGE -> +inf (plan7.pyx:5012-5013)
            return math.inf
        if st != 0:
            raise status_to_exception(st, fn.__name__, _lib.last_error())
        return float(sc.value)

    def msv_filter(self, seq: DigitalSequence, device: int = 0) -> float:
        return self._one(_lib.lib().p7x_msv_filter, seq, device)

    def ssv_filter(self, seq: DigitalSequence, device: int = 0) -> float:
        # p7_SSVFilter returns the MSV score whenever it returns eslOK (plan7.pyx:5033-5038)
        return self._one(_lib.lib().p7x_msv_filter, seq, device)

    def viterbi_filter(self, seq: DigitalSequence, device: int = 0) -> float:
        return self._one(_lib.lib().p7x_vit_filter, seq, device)

    def forward_parser(self, seq: DigitalSequence, device: int = 0) -> float:
        return self._one(_lib.lib().p7x_fwd_parser, seq, device)

    def backward_parser(self, seq: DigitalSequence, device: int = 0) -> float:
        return self._one(_lib.lib().p7x_bck_parser, seq, device)


class Profile:
    """A configured generic profile (reference ``plan7.pyx:7767-8267``).  Configuration and
    conversion are one native call here, so this object only records the arguments of
    ``configure`` and defers to :class:`OptimizedProfile`."""

    def __init__(self, M: int, alphabet: Alphabet):
        self.alphabet = alphabet
        self.M = M
        self._hmm: Optional[HMM] = None
        self._bg: Optional[Background] = None
        self.L = 400

    def configure(self, hmm: HMM, background: Background, L: int = 400, multihit: bool = True, local: bool = True):
        if hmm.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, hmm.alphabet)
        if background.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, background.alphabet)
        if not (multihit and local):
            raise InvalidParameter("multihit/local", (multihit, local), hint="only local multihit mode is on the search path")
        self._hmm, self._bg, self.L, self.M = hmm, background, int(L), hmm.M
        self.name, self.accession, self.description = hmm.name, hmm.accession, hmm.description

    def to_optimized(self) -> OptimizedProfile:
        if self._hmm is None:
            raise ValueError("profile is not configured")
        return OptimizedProfile(self._hmm, self._bg, self.L)
