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
import sys
import weakref
from typing import Iterable, Iterator, List, Optional, Sequence, Union

import itertools
import threading

import numpy as np

from . import _lib
from .easel import (Alphabet, DigitalSequence, DigitalSequenceBlock, SequenceFile, eslAMINO, eslDNA,
                    eslRNA)
from .errors import (AlphabetMismatch, AllocationError, InvalidParameter, MissingCutoffs,
                     UnexpectedError, status_to_exception)

__all__ = [
    "HMM", "HMMFile", "HMMPressedFile", "Background", "Profile", "OptimizedProfile", "OptimizedProfileBlock", "EvalueParameters", "Cutoffs",
    "Pipeline", "LongTargetsPipeline", "SequenceDatabase", "TopHits", "Hit", "Domain", "Domains", "Alignment",
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

    def __setattr__(self, name, value):
        """The parameters can be set (``None`` clears one), as in the reference (``plan7.pyx:1760-1849``); they are the HMM's own.
        An `OptimizedProfile` hands out a read-only copy (its values live in the search-ready C object): setting raises."""
        if name in self._names:
            if not self._v.flags.writeable:
                raise AttributeError(f"the E-value parameters of an OptimizedProfile are read-only (set them on the HMM): {name}")
            self._v[self._names.index(name)] = EVPARAM_UNSET if value is None else value
        else:
            object.__setattr__(self, name, value)

    def as_vector(self) -> np.ndarray:
        return self._v.copy()

    def __eq__(self, other):
        if isinstance(other, EvalueParameters):
            return bool(np.array_equal(self._v, other._v))
        return NotImplemented

    __hash__ = None

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

    def _set_pair(self, i, pair):
        if not self._v.flags.writeable:      # an OptimizedProfile's copy: a write here would be silently dropped
            raise AttributeError("the cutoffs of an OptimizedProfile are read-only (set them on the HMM before optimizing it)")
        if pair is None:
            self._v[i] = self._v[i + 1] = CUTOFF_UNSET
        else:
            a, b = pair
            self._v[i], self._v[i + 1] = a, b

    # each pair can be read, set from two floats, and cleared with None or ``del`` (reference ``plan7.pyx:1204-1420``); the values
    # are the owner's (``HMM.cutoffs`` hands out a view; ``OptimizedProfile.cutoffs`` a READ-ONLY copy: setting raises)
    gathering = property(lambda self: self._pair(0), lambda self, v: self._set_pair(0, v), lambda self: self._set_pair(0, None))
    trusted = property(lambda self: self._pair(2), lambda self, v: self._set_pair(2, v), lambda self: self._set_pair(2, None))
    noise = property(lambda self: self._pair(4), lambda self, v: self._set_pair(4, v), lambda self: self._set_pair(4, None))
    gathering1 = property(lambda self: None if self._pair(0) is None else self._pair(0)[0])
    gathering2 = property(lambda self: None if self._pair(0) is None else self._pair(0)[1])
    trusted1 = property(lambda self: None if self._pair(2) is None else self._pair(2)[0])
    trusted2 = property(lambda self: None if self._pair(2) is None else self._pair(2)[1])
    noise1 = property(lambda self: None if self._pair(4) is None else self._pair(4)[0])
    noise2 = property(lambda self: None if self._pair(4) is None else self._pair(4)[1])

    def __eq__(self, other):
        if isinstance(other, Cutoffs):
            return bool(np.array_equal(self._v, other._v))
        return NotImplemented

    __hash__ = None

    def __repr__(self):
        return f"<Cutoffs gathering={self.gathering!r} trusted={self.trusted!r} noise={self.noise!r}>"

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

    # ---- copies, comparison, pickling (reference plan7.pyx:2418-2560, 3124-3140)
    _STATE = ("accession", "description", "composition", "consensus", "consensus_structure", "reference", "model_mask", "map", "nseq",
              "nseq_effective", "max_length", "checksum", "command_line", "creation_time")

    def copy(self) -> "HMM":
        new = HMM(self.alphabet, self.M, self.name)
        new.transition_probabilities[:] = self.transition_probabilities
        new.match_emissions[:] = self.match_emissions
        new.insert_emissions[:] = self.insert_emissions
        new._evparam[:] = self._evparam
        new._cutoffs[:] = self._cutoffs
        for attr in self._STATE:
            v = getattr(self, attr)
            setattr(new, attr, v.copy() if isinstance(v, np.ndarray) else v)
        return new

    def __copy__(self) -> "HMM":
        return self.copy()

    def __deepcopy__(self, memo) -> "HMM":
        if id(self) not in memo:
            memo[id(self)] = self.copy()
        return memo[id(self)]

    def __eq__(self, other):
        """``p7_hmm_Compare`` with a tolerance of zero: alphabet, size, every parameter, every annotation."""
        if not isinstance(other, HMM):
            return NotImplemented
        if self is other:
            return True
        if self.alphabet != other.alphabet or self.M != other.M or self.name != other.name:
            return False
        for a, b in ((self.transition_probabilities, other.transition_probabilities), (self.match_emissions, other.match_emissions),
                     (self.insert_emissions, other.insert_emissions), (self._evparam, other._evparam), (self._cutoffs, other._cutoffs)):
            if not np.array_equal(a, b):
                return False
        for attr in self._STATE:
            a, b = getattr(self, attr), getattr(other, attr)
            if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
                if a is None or b is None or not np.array_equal(a, b):
                    return False
            elif a != b:
                return False
        return True

    __hash__ = None

    def __reduce__(self):
        return HMM, (self.alphabet, self.M, self.name), self.__getstate__()

    def __getstate__(self) -> dict:
        state = {attr: getattr(self, attr) for attr in self._STATE}
        state.update(t=self.transition_probabilities, mat=self.match_emissions, ins=self.insert_emissions, evparam=self._evparam,
                     cutoff=self._cutoffs)
        return state

    def __setstate__(self, state: dict) -> None:
        self.transition_probabilities[:] = state["t"]
        self.match_emissions[:] = state["mat"]
        self.insert_emissions[:] = state["ins"]
        self._evparam[:] = state["evparam"]
        self._cutoffs[:] = state["cutoff"]
        for attr in self._STATE:
            setattr(self, attr, state.get(attr))

    # ---- model statistics and edits (reference plan7.pyx:3335-3402, 3459-3520, 3591-3655; upstream p7_hmm.c, modelstats.c)
    def _occupancy(self):
        """``p7_hmm_CalculateOccupancy``: the probability that a glocal path uses match state k, and insert state k."""
        t = self.transition_probabilities.astype(np.float32)
        M = self.M
        mocc = np.zeros(M + 1, dtype=np.float32)
        iocc = np.zeros(M + 1, dtype=np.float32)
        if M >= 1:
            mocc[1] = t[0, 1] + t[0, 0]
            for k in range(2, M + 1):
                mocc[k] = mocc[k - 1] * (t[k - 1, 0] + t[k - 1, 1]) + (np.float32(1.0) - mocc[k - 1]) * t[k - 1, 5]
        with np.errstate(divide="ignore", invalid="ignore"):
            iocc[0] = t[0, 1] / t[0, 3]
            iocc[1:] = mocc[1:] * t[1:, 1] / t[1:, 3]
        return mocc, iocc

    def match_occupancy(self) -> np.ndarray:
        return self._occupancy()[0]

    def mean_match_entropy(self) -> float:
        """Mean entropy of the match emission distributions, in bits (``p7_MeanMatchEntropy``)."""
        p = self.match_emissions[1:].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            h = -np.where(p > 0, p * np.log2(p), 0.0).sum(axis=1)
        return float(h.mean())

    def mean_match_information(self, background: "Background") -> float:
        """Mean information content of the match states against the null model, in bits (``p7_MeanMatchInfo``)."""
        f = background.residue_frequencies.astype(np.float64)
        hbg = -float(np.where(f > 0, f * np.log2(f), 0.0).sum())
        return hbg - self.mean_match_entropy()

    def mean_match_relative_entropy(self, background: "Background") -> float:
        """Mean relative entropy of the match states against the null model, in bits (``p7_MeanMatchRelativeEntropy``)."""
        p = self.match_emissions[1:].astype(np.float64)
        f = background.residue_frequencies.astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            d = np.where(p > 0, p * np.log2(p / f), 0.0).sum(axis=1)
        return float(d.mean())

    def renormalize(self) -> None:
        """``p7_hmm_Renormalize``: every emission row and the three transition distributions of every node sum to one; node M keeps
        its conventions (D -> M_end is 1, no M -> D)."""
        def norm(a):
            tot = a.sum(axis=1, keepdims=True)
            n = a.shape[1]
            a[:] = np.where(tot > 0, a / np.where(tot > 0, tot, 1), np.float32(1.0 / n))       # esl_vec_FNorm: a zero vector becomes uniform
        norm(self.match_emissions)
        norm(self.insert_emissions)
        t = self.transition_probabilities
        norm(t[:, 0:3]); norm(t[:, 3:5]); norm(t[:, 5:7])
        M = self.M
        t[M, 5], t[M, 6] = 1.0, 0.0
        if t[M, 2] > 0.0:
            t[M, 2] = 0.0; t[M, 0] = 0.5; t[M, 1] = 0.5

    def scale(self, scale: float, exponential: bool = False) -> None:
        """Rescale a model that holds counts (``p7_hmm_Scale`` / ``p7_hmm_ScaleExponential``: there the factor of node k is
        count_k ** scale / count_k, count_k the node's match emission counts)."""
        if not exponential:
            for a in (self.transition_probabilities, self.match_emissions, self.insert_emissions):
                a *= np.float32(scale)
            return
        for k in range(1, self.M + 1):
            count = float(self.match_emissions[k].sum())
            f = np.float32((count ** scale) / count) if count > 0 else np.float32(1.0)
            self.transition_probabilities[k] *= f
            self.match_emissions[k] *= f
            self.insert_emissions[k] *= f

    def zero(self) -> None:
        """Set every parameter to zero, the composition included (``p7_hmm_Zero``); annotation stays."""
        self.transition_probabilities[:] = 0
        self.match_emissions[:] = 0
        self.insert_emissions[:] = 0
        if self.composition is not None:
            self.composition[:] = 0

    def set_composition(self) -> None:
        """``p7_hmm_SetComposition``: the expected residue composition of a sequence emitted by the model -- the emissions of its
        match and insert states weighted by their occupancies."""
        mocc, iocc = self._occupancy()
        with np.errstate(invalid="ignore"):
            compo = np.nan_to_num(iocc[0]) * self.insert_emissions[0].astype(np.float32)
            for k in range(1, self.M + 1):
                compo = compo + mocc[k] * self.match_emissions[k] + np.nan_to_num(iocc[k]) * self.insert_emissions[k]
        tot = float(compo.sum())
        self.composition = (compo / tot if tot > 0 else np.full(self.alphabet.K, 1.0 / self.alphabet.K)).astype(np.float32)

    def set_consensus(self, sequence=None) -> None:
        """``p7_hmm_SetConsensus``: the residue of highest emission probability at every node, upper case when that probability
        reaches 0.5 (amino) / 0.9 (nucleic); with a digital ``sequence`` of M residues (a single-sequence model), that sequence."""
        if sequence is None:
            self.consensus = _set_consensus(self)
            return
        if sequence.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, sequence.alphabet)
        dsq = np.asarray(sequence.sequence)
        if dsq.shape[0] < self.M:
            raise ValueError(f"Expected `DigitalSequence` of length {self.M!r}, found {dsq.shape[0]!r}")
        thresh = 0.5 if self.alphabet.is_amino() else 0.9
        out = []
        for k in range(1, self.M + 1):
            x = int(dsq[k - 1])
            c = self.alphabet.symbols[x]
            out.append(c.upper() if x < self.alphabet.K and self.match_emissions[k, x] >= thresh else c.lower())
        self.consensus = "".join(out)

    def to_profile(self, background: Optional["Background"] = None, L: int = 400, multihit: bool = True, local: bool = True) -> "Profile":
        profile = Profile(self.M, self.alphabet)
        profile.configure(self, Background(self.alphabet) if background is None else background, L=L, multihit=multihit, local=local)
        return profile

    def validate(self, tolerance: float = 1e-4) -> None:
        """``p7_hmm_Validate``: the structural constraints of a Plan7 model; ``ValueError`` names the first that fails."""
        t, M = self.transition_probabilities, self.M
        if M < 1:
            raise ValueError("HMM has M < 1")
        def check(a, what, first=0):
            bad = np.nonzero(np.abs(a.sum(axis=1) - 1.0) > tolerance)[0]
            if bad.size:
                raise ValueError(f"Invalid HMM: {what} of node {int(bad[0]) + first} does not sum to 1")
        if abs(float(self.match_emissions[0, 0]) - 1.0) > tolerance and float(self.match_emissions[0].sum()) != 0.0:
            raise ValueError("Invalid HMM: mat[0] is not the unused-node convention (1, 0, ...)")
        check(self.match_emissions[1:], "match emissions", 1)
        check(self.insert_emissions, "insert emissions")
        check(t[:, 0:3], "match transitions")
        check(t[:, 3:5], "insert transitions")
        check(t[1:, 5:7], "delete transitions", 1)
        if t[M, 2] != 0.0 or t[M, 6] != 0.0:
            raise ValueError("Invalid HMM: node M must not enter a delete state")
        for attr in ("consensus", "consensus_structure", "reference", "model_mask"):
            v = getattr(self, attr)
            if v is not None and len(v) != M:
                raise ValueError(f"Invalid HMM: {attr} annotation has {len(v)} characters for {M} nodes")

    def _flags(self) -> int:
        """The P7_HMM flag word a save file carries: which optional fields this model holds."""
        F = HMMFile._F
        unset = lambda x: float(x) == CUTOFF_UNSET
        have = dict(DESC=self.description is not None, RF=self.reference is not None, CS=self.consensus_structure is not None,
                    STATS=float(self._evparam[0]) != EVPARAM_UNSET, MAP=self.map is not None, ACC=self.accession is not None,
                    GA=not unset(self._cutoffs[0]), TC=not unset(self._cutoffs[2]), NC=not unset(self._cutoffs[4]),
                    COMPO=self.composition is not None, CHKSUM=self.checksum is not None, CONS=self.consensus is not None,
                    MMASK=self.model_mask is not None)
        return sum(F[k] for k, v in have.items() if v)

    def write(self, fh, binary: bool = False) -> None:
        """Write the model to a file object (reference ``plan7.pyx:3403-3436``, upstream ``p7_hmmfile_WriteASCII`` /
        ``p7_hmmfile_WriteBinary``): the HMMER3/f text format to a text or a binary handle, or the 3/f binary format
        (``binary=True``, what ``hmmpress`` keeps in ``.h3m``) to a binary handle."""
        if binary:
            fh.write(self._to_binary())
            return
        text = self._to_text()
        try:
            fh.write(text)
        except TypeError:
            fh.write(text.encode())

    def _to_binary(self) -> bytes:
        import struct
        M, K = self.M, self.alphabet.K
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).tobytes()

        def string(s):
            if s is None:
                return struct.pack("=i", 0)
            b = s.encode() + b"\0"
            return struct.pack("=i", len(b)) + b

        def line(s):
            return (" " + s).encode()[:M + 1].ljust(M + 1, b" ") + b"\0"

        flags = self._flags()
        F = HMMFile._F
        out = [struct.pack("=Iiii", HMMFile._MAGIC_3F, flags, M, self.alphabet.type_code),
               f32(self.match_emissions[1:]), f32(self.insert_emissions), f32(self.transition_probabilities), string(self.name)]
        if flags & F["ACC"]:
            out.append(string(self.accession))
        if flags & F["DESC"]:
            out.append(string(self.description))
        for key, s in (("RF", self.reference), ("MMASK", self.model_mask), ("CONS", self.consensus), ("CS", self.consensus_structure)):
            if flags & F[key]:
                out.append(line(s))
        out.append(string(self.command_line))
        out.append(struct.pack("=ifi", -1 if self.nseq is None else int(self.nseq), -1.0 if self.nseq_effective is None else float(self.nseq_effective),
                               -1 if self.max_length is None else int(self.max_length)))
        out.append(string(self.creation_time))
        if flags & F["MAP"]:
            out.append(np.ascontiguousarray(self.map, dtype=np.int32).tobytes())
        out.append(struct.pack("=I", int(self.checksum or 0)))
        out.append(f32(self._evparam) + f32(self._cutoffs))
        if flags & F["COMPO"]:
            out.append(f32(self.composition))
        return b"".join(out)

    def _to_text(self) -> str:
        M, abc = self.M, self.alphabet

        def probs(p):
            p = np.asarray(p, dtype=np.float32)
            with np.errstate(divide="ignore"):
                nl = -np.log(p)
            return "".join("        *" if x == 0.0 else (" %8.5f" % (0.0 if x == 1.0 else v)) for x, v in zip(p.tolist(), nl.tolist()))

        yn = lambda x: "yes" if x is not None else "no"
        o = ["HMMER3/f [pyhmmer_amd | 3.4 layout]\n", f"NAME  {self.name}\n"]
        if self.accession is not None:
            o.append(f"ACC   {self.accession}\n")
        if self.description is not None:
            o.append(f"DESC  {self.description}\n")
        o.append(f"LENG  {M}\n")
        if self.max_length is not None and self.max_length > 0:
            o.append(f"MAXL  {self.max_length}\n")
        o.append("ALPH  %s\n" % {eslAMINO: "amino", eslDNA: "DNA", eslRNA: "RNA"}[abc.type_code])
        o.append(f"RF    {yn(self.reference)}\nMM    {yn(self.model_mask)}\nCONS  {yn(self.consensus)}\nCS    {yn(self.consensus_structure)}\nMAP   {yn(self.map)}\n")
        if self.creation_time is not None:
            o.append(f"DATE  {self.creation_time}\n")
        if self.command_line is not None:
            o.extend(f"COM   [{i}] {c}\n" for i, c in enumerate(self.command_line.split("\n"), 1))
        if self.nseq is not None and self.nseq > 0:
            o.append(f"NSEQ  {self.nseq}\n")
        if self.nseq_effective is not None and self.nseq_effective >= 0:
            o.append("EFFN  %f\n" % self.nseq_effective)
        if self.checksum is not None:
            o.append(f"CKSUM {self.checksum}\n")
        for key, idx in (("GA", 0), ("TC", 2), ("NC", 4)):
            if float(self._cutoffs[idx]) != CUTOFF_UNSET:
                o.append("%-5s %.2f %.2f\n" % (key, self._cutoffs[idx], self._cutoffs[idx + 1]))
        if float(self._evparam[0]) != EVPARAM_UNSET:
            for label, idx in (("MSV", 0), ("VITERBI", 2), ("FORWARD", 4)):
                o.append("STATS LOCAL %-8s %8.4f %8.5f\n" % (label, self._evparam[idx], self._evparam[idx + 1]))
        o.append("HMM     " + "".join(f"     {c}   " for c in abc.symbols[:abc.K]) + "\n")
        o.append("       " + "".join(" %8s" % s for s in ("m->m", "m->i", "m->d", "i->m", "i->i", "d->m", "d->d")) + "\n")
        if self.composition is not None:
            o.append("  COMPO " + probs(self.composition) + "\n")
        for k in range(M + 1):
            if k > 0:
                ann = lambda s: "-" if s is None else s[k - 1]
                o.append(" %6d " % k + probs(self.match_emissions[k]) + (" %6d" % self.map[k] if self.map is not None else " %6s" % "-")
                         + f" {ann(self.consensus)} {ann(self.reference)} {ann(self.model_mask)} {ann(self.consensus_structure)}\n")
            o.append("        " + probs(self.insert_emissions[k]) + "\n")
            o.append("        " + probs(self.transition_probabilities[k]) + "\n")
        o.append("//\n")
        return "".join(o)

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
    """Reader for HMMER3 save files (reference ``plan7.pyx:3656-4050``): the ASCII format (upstream
    ``p7_hmmfile.c:read_asc30hmm``) and the binary ``.h3m`` format 3/f (``read_bin30hmm``; the reference tells them apart by the
    four magic bytes, ``plan7.pyx:3712-3760``).  Pressed ``.h3f/.h3p`` databases: ``HMMPressedFile``."""

    _MAGIC_3F = 0xe8ededba            # "hmma" + 0x80808080 (patches/p7_hmmfile.c.patch:15-20); older binary versions are not read
    _MAGIC_OLD = (0xe8ededb6, 0xe8ededb7, 0xe8ededb8, 0xe8ededb9, 0xe8ededb0)
    # P7_HMM flags (upstream p7_hmm.h) that say which optional fields a binary record holds
    _F = dict(DESC=1 << 1, RF=1 << 2, CS=1 << 3, STATS=1 << 7, MAP=1 << 8, ACC=1 << 9, GA=1 << 10, TC=1 << 11, NC=1 << 12, CA=1 << 13,
              COMPO=1 << 14, CHKSUM=1 << 15, CONS=1 << 16, MMASK=1 << 17)

    def __init__(self, file, db: bool = True, *, alphabet: Optional[Alphabet] = None):
        """``file``: a path or a file object.  ``db`` (reference ``plan7.pyx:3697-3706``, ``p7_hmmfile_Open`` / ``_OpenNoDB``):
        with a path, a pressed database beside it is used if there is one -- the models are then read from ``<path>.h3m`` and
        ``optimized_profiles()`` walks ``<path>.h3f`` / ``<path>.h3p``; ``db=False`` ignores it."""
        self._binary = False
        self._closed = False
        self._pressed = None
        if isinstance(file, (str, bytes, os.PathLike)):
            base = os.fsdecode(os.fspath(file))
            if db and os.path.isfile(base + ".h3m"):           # upstream looks for the pressed database first
                self._pressed = base if os.path.isfile(base + ".h3f") and os.path.isfile(base + ".h3p") else None
                file = base + ".h3m"
            if os.path.isdir(file):
                raise IsADirectoryError(21, f"Is a directory: {base!r}")
            if not os.path.exists(file):
                raise FileNotFoundError(2, f"No such file or directory: {base!r}")
            if os.stat(file).st_size == 0:
                raise EOFError("HMM file is empty")
            with open(file, "rb") as probe:
                first = probe.read(256)
            head = first[:4]
            if not first.lstrip().startswith(b"HMMER") and (len(head) < 4 or int.from_bytes(head, sys.byteorder) not in (self._MAGIC_3F,) + self._MAGIC_OLD):
                raise ValueError("format not recognized by HMMER")
            magic = int.from_bytes(head, sys.byteorder) if len(head) == 4 else 0
            self._binary = magic == self._MAGIC_3F
            if magic in self._MAGIC_OLD:
                raise ValueError(f"{os.fspath(file)!r}: binary HMM file of an older format (HMMER 3/a-3/e); only 3/f is read")
            self._fh = open(file, "rb" if self._binary else "r")
            self._own = True
            self.name = base
        else:
            self._fh = file
            self._own = False
            self.name = getattr(file, "name", None)
            head = None
            if hasattr(file, "peek"):
                head = file.peek(4)[:4]
            elif hasattr(file, "seek") and hasattr(file, "tell"):
                at = file.tell(); head = file.read(4); file.seek(at)
            if isinstance(head, (bytes, bytearray)) and len(head) == 4:
                magic = int.from_bytes(head, sys.byteorder)
                self._binary = magic == self._MAGIC_3F
                if magic in self._MAGIC_OLD:
                    raise ValueError(f"{self.name!r}: binary HMM file of an older format (HMMER 3/a-3/e); only 3/f is read")
            if not self._binary and (isinstance(head, (bytes, bytearray)) or isinstance(getattr(file, "read", lambda n: "")(0), (bytes, bytearray))):
                # a text-format HMM behind a BINARY handle (open(p, "rb"), BytesIO: what the reference's HMMFile takes): the
                # text parser reads str lines
                import io
                self._fh = io.TextIOWrapper(file, encoding="ascii", errors="replace", newline=None)
        self._alphabet = alphabet

    def _read_binary(self) -> Optional[HMM]:
        """One record of the 3/f binary format, as upstream's write_bin_hmm lays it out: magic, flags, M, alphabet type; the
        match emissions of nodes 1..M, the insert emissions and the transitions of nodes 0..M (float32 PROBABILITIES, not the
        text format's negative logarithms); name [accession] [description] as length-prefixed strings; the annotation lines
        (M + 2 bytes each: a blank, the M characters, NUL) that the flags announce; command line, nseq, effective nseq,
        max_length, date, [map], checksum, E-value parameters, cutoffs, [composition]."""
        import struct
        fh = self._fh
        head = fh.read(4)
        if len(head) == 0:
            return None
        if len(head) < 4 or int.from_bytes(head, sys.byteorder) != self._MAGIC_3F:
            raise ValueError(f"Invalid format in file: {self.name!r} (bad magic in a binary HMM record)")

        def take(n):
            b = fh.read(n)
            if len(b) != n:
                raise ValueError(f"premature end of binary HMM file {self.name!r}")
            return b

        def word(fmt):
            return struct.unpack("=" + fmt, take(struct.calcsize("=" + fmt)))

        def string():
            n, = word("i")
            if n < 0 or n > (1 << 24):
                raise ValueError(f"corrupt string length in binary HMM file {self.name!r}")
            return None if n == 0 else take(n)[:-1].decode()

        flags, M, abc_type = word("iii")
        alphabet = {eslAMINO: Alphabet.amino, eslDNA: Alphabet.dna, eslRNA: Alphabet.rna}.get(abc_type)
        if alphabet is None or M < 1 or M > 100000:
            raise ValueError(f"corrupt header in binary HMM file {self.name!r}")
        alphabet = alphabet()
        if self._alphabet is not None and self._alphabet != alphabet:
            raise AlphabetMismatch(self._alphabet, alphabet)
        K, F = alphabet.K, self._F
        mat = np.frombuffer(take(4 * M * K), dtype=np.float32).reshape(M, K)
        ins = np.frombuffer(take(4 * (M + 1) * K), dtype=np.float32).reshape(M + 1, K)
        t = np.frombuffer(take(4 * (M + 1) * 7), dtype=np.float32).reshape(M + 1, 7)
        hmm = HMM(alphabet, M, string() or "")
        hmm.match_emissions[1:] = mat
        hmm.match_emissions[0, 0] = 1.0
        hmm.insert_emissions[:] = ins
        hmm.transition_probabilities[:] = t
        if flags & F["ACC"]:
            hmm.accession = string()
        if flags & F["DESC"]:
            hmm.description = string()
        line = lambda: take(M + 2)[1:M + 1].decode()
        if flags & F["RF"]:
            hmm.reference = line()
        if flags & F["MMASK"]:
            hmm.model_mask = line()
        if flags & F["CONS"]:
            hmm.consensus = line()
        if flags & F["CS"]:
            hmm.consensus_structure = line()
        if flags & F["CA"]:
            line()                                   # surface accessibility: not kept by the text reader either
        hmm.command_line = string()
        hmm.nseq, = word("i")
        eff, = word("f")
        hmm.nseq_effective = float(eff)
        maxl, = word("i")
        if maxl > 0:
            hmm.max_length = int(maxl)
        hmm.creation_time = string()
        if flags & F["MAP"]:
            hmm.map = np.array(word(f"{M + 1}i"), dtype=np.int64)
        cks, = word("I")
        if flags & F["CHKSUM"]:
            hmm.checksum = int(cks)
        ev = word("6f")
        if flags & F["STATS"]:
            hmm._evparam[:] = ev
        cut = word("6f")
        for key, idx in (("GA", 0), ("TC", 2), ("NC", 4)):
            if flags & F[key]:
                hmm._cutoffs[idx], hmm._cutoffs[idx + 1] = cut[idx], cut[idx + 1]
        if flags & F["COMPO"]:
            hmm.composition = np.array(word(f"{K}f"), dtype=np.float32)
        if hmm.nseq is not None and hmm.nseq < 0:
            hmm.nseq = None
        if hmm.nseq_effective is not None and hmm.nseq_effective < 0:
            hmm.nseq_effective = None
        if hmm.consensus is None:
            hmm.consensus = _set_consensus(hmm)
        return hmm

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._own and not self._closed:
            self._fh.close()
        self._closed = True

    @property
    def closed(self) -> bool:
        """Whether the file is closed (reference ``plan7.pyx:3903-3907``)."""
        return self._closed

    def __repr__(self):
        return f"{type(self).__name__}({self.name!r})" if self.name is not None else f"<{type(self).__name__} file={self._fh!r}>"

    def is_pressed(self) -> bool:
        """Whether the file is a pressed HMM database (reference ``plan7.pyx:4009-4029``): ``<path>.h3m``, ``.h3f`` and ``.h3p``
        are there (the ``.h3i`` index of upstream's ``hmmpress`` is not needed by anything on this path)."""
        if self._closed:
            raise ValueError("I/O operation on closed file.")
        return self._pressed is not None

    def optimized_profiles(self) -> "HMMPressedFile":
        """An iterator over the optimized profiles of the pressed database (reference ``plan7.pyx:4031-4048``)."""
        if self._closed:
            raise ValueError("I/O operation on closed file.")
        if self._pressed is None:
            raise ValueError("HMM file does not contain optimized profiles.")
        return HMMPressedFile(self._pressed, alphabet=self._alphabet)

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
        if self._closed:
            raise ValueError("I/O operation on closed file.")
        if self._binary:
            return self._read_binary()
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
            elif tag == "COM":            # "COM   [n] command": upstream drops the counter and joins the lines with newlines
                cmd = val
                if cmd.startswith("[") and "]" in cmd:
                    cmd = cmd[cmd.index("]") + 1:].strip()
                hdr["COM"] = (hdr["COM"] + "\n" + cmd) if "COM" in hdr else cmd
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
            hmm.nseq_effective = float(np.float32(hdr["EFFN"]))      # upstream holds a float: the text and the binary form of a model then agree
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

    @classmethod
    def _from_handle(cls, alphabet: Alphabet, handle) -> "OptimizedProfile":
        """Wrap a ``p7x_oprofile`` that was built elsewhere (a pressed record)."""
        self = cls.__new__(cls)
        self.alphabet = alphabet
        self._handle = handle
        self._info = _lib.OprofileInfo()
        _lib.lib().p7x_oprofile_get_info(self._handle, C.byref(self._info))
        self._hmm = None
        return self

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.lib().p7x_oprofile_destroy(h)
            except Exception:
                pass
            self._handle = None

    def write(self, fh_filter, fh_profile, offsets=None) -> None:
        """Write the profile in pressed form: the MSV part to ``fh_filter`` (``.h3f``) and the rest to ``fh_profile``
        (``.h3p``), both binary file objects (reference ``OptimizedProfile.write``, ``plan7.pyx:5078-5105``).
        ``offsets``: byte offsets of this model in the ``.h3m``, ``.h3f`` and ``.h3p`` files of a pressed
        database (stored inside the ``.h3f`` record); default: the current positions of the two files and 0."""
        lf, lp = C.c_size_t(), C.c_size_t()
        st = _lib.lib().p7x_oprofile_write_pressed(self._handle, None, None, 0, C.byref(lf), None, 0, C.byref(lp))
        if st != 0:
            raise status_to_exception(st, "p7x_oprofile_write_pressed", _lib.last_error())
        if offsets is None:
            offsets = (0, fh_filter.tell(), fh_profile.tell())
        offs = (C.c_int64 * 3)(*[int(o) for o in offsets])
        bf, bp = (C.c_uint8 * lf.value)(), (C.c_uint8 * lp.value)()
        st = _lib.lib().p7x_oprofile_write_pressed(self._handle, offs, bf, lf.value, C.byref(lf), bp, lp.value, C.byref(lp))
        if st != 0:
            raise status_to_exception(st, "p7x_oprofile_write_pressed", _lib.last_error())
        fh_filter.write(bytes(bf))
        fh_profile.write(bytes(bp))

    def _fill_names(self) -> None:
        buf = C.create_string_buffer(1 << 16)
        vals = []
        for which in range(4):
            n = _lib.lib().p7x_oprofile_get_string(self._handle, which, buf, len(buf))
            vals.append(buf.value.decode() if n > 0 else None)
        self.name, self.accession, self.description = vals[0], vals[1], vals[2]
        self.consensus = vals[3][1:] if vals[3] else None

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

    @staticmethod
    def _frozen(values) -> np.ndarray:
        a = np.asarray(values, dtype=np.float32)
        a.flags.writeable = False
        return a

    @property
    def evalue_parameters(self) -> EvalueParameters:
        return EvalueParameters(self._frozen([self._info.evparam[i] for i in range(6)]))

    @property
    def cutoffs(self) -> Cutoffs:
        return Cutoffs(self._frozen([self._info.cutoff[i] for i in range(6)]))

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


_P7X_SEARCH_SEQS, _P7X_SCAN_MODELS = 0, 1      # p7x.h / p7_pipeline.pxd:26-28


class OptimizedProfileBlock:
    """A container of `OptimizedProfile` objects for the scan loop (reference ``plan7.pyx:5072-5338``: a Python list
    kept in step with a C array of ``P7_OPROFILE*`` plus one lock per profile, because ``_scan_loop`` mutates the
    length model of every profile it visits).  Here the profiles are never mutated by a search -- the length model
    is a per-target table lookup on the device -- so the block is a typed list; ``Pipeline.scan_seq`` and
    ``hmmer.hmmscan`` hand its members to the device in batches (``p7x_search_batch_enqueue``) and a block whose
    profiles have been searched once keeps their device images resident in HBM."""

    def __init__(self, alphabet: Alphabet, iterable=()):
        self.alphabet = alphabet
        self._storage: List[OptimizedProfile] = []
        self.extend(iterable)

    def _check(self, om) -> "OptimizedProfile":
        if not isinstance(om, OptimizedProfile):
            raise TypeError(f"Expected OptimizedProfile, found {type(om).__name__}")
        if om.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, om.alphabet)
        return om

    def __len__(self) -> int:
        return len(self._storage)

    def __contains__(self, item) -> bool:
        return isinstance(item, OptimizedProfile) and any(item is om for om in self._storage)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return OptimizedProfileBlock(self.alphabet, self._storage[index])
        return self._storage[index]

    def __setitem__(self, index, value) -> None:
        if isinstance(index, slice):
            self._storage[index] = [self._check(om) for om in value]
        else:
            self._storage[index] = self._check(value)

    def __delitem__(self, index) -> None:
        del self._storage[index]

    def __iter__(self):
        return iter(self._storage)

    def __eq__(self, other) -> bool:
        if not isinstance(other, OptimizedProfileBlock):
            return NotImplemented
        return self.alphabet == other.alphabet and len(self) == len(other) and all(a is b for a, b in zip(self, other))

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self.alphabet!r}, {self._storage!r})"

    def __copy__(self) -> "OptimizedProfileBlock":
        return self.copy()

    def append(self, optimized_profile) -> None:
        self._storage.append(self._check(optimized_profile))

    def clear(self) -> None:
        self._storage.clear()

    def extend(self, iterable) -> None:
        for om in iterable:
            self.append(om)

    def index(self, optimized_profile, start: int = 0, stop: int = sys.maxsize) -> int:
        for i in range(*slice(start, stop).indices(len(self._storage))):
            if self._storage[i] is optimized_profile:
                return i
        raise ValueError(f"{optimized_profile!r} is not in block")

    def insert(self, index: int, optimized_profile) -> None:
        self._storage.insert(index, self._check(optimized_profile))

    def pop(self, index: int = -1) -> "OptimizedProfile":
        return self._storage.pop(index)

    def remove(self, optimized_profile) -> None:
        del self._storage[self.index(optimized_profile)]

    def copy(self) -> "OptimizedProfileBlock":
        return OptimizedProfileBlock(self.alphabet, self._storage)


class HMMPressedFile:
    """Iterator over the optimized profiles of a pressed HMM database: ``<path>.h3f`` + ``<path>.h3p`` as written by
    ``hmmpress`` (reference ``plan7.pyx:4051-4197``; the ``.h3m`` / ``.h3i`` companions are not needed to search).
    Yields :class:`OptimizedProfile` objects whose tables are exactly the stored ones."""

    def __init__(self, path, alphabet: Optional[Alphabet] = None):
        base = os.fspath(path)
        for ext in (".h3f", ".h3p"):
            if not os.path.exists(base + ext):
                raise FileNotFoundError(2, f"no such file or directory: {base + ext!r}")
        self.name = base
        self._f = np.fromfile(base + ".h3f", dtype=np.uint8)
        self._p = np.fromfile(base + ".h3p", dtype=np.uint8)
        self._pf = self._pp = 0
        self._alphabet = alphabet
        self._bg = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self) -> None:
        self._f = self._p = np.zeros(0, dtype=np.uint8)

    def rewind(self) -> None:
        self._pf = self._pp = 0

    def __iter__(self):
        return self

    def __len__(self) -> int:
        pf, pp = self._pf, self._pp
        self.rewind()
        n = sum(1 for _ in self)
        self._pf, self._pp = pf, pp
        return n

    def __next__(self) -> "OptimizedProfile":
        if self._pf >= self._f.shape[0]:
            raise StopIteration
        if self._f.shape[0] - self._pf < 12:
            raise ValueError(f"{self.name}.h3f: truncated record")
        abc_type = int(self._f[self._pf + 8:self._pf + 12].view(np.int32)[0])
        abc = {3: Alphabet.amino, 2: Alphabet.dna, 1: Alphabet.rna}.get(abc_type)
        if abc is None:
            raise ValueError(f"{self.name}.h3f: unknown alphabet type {abc_type}")
        alphabet = abc()
        if self._alphabet is not None and self._alphabet != alphabet:
            raise AlphabetMismatch(self._alphabet, alphabet)
        bg = self._bg.setdefault(abc_type, np.ascontiguousarray(Background(alphabet).residue_frequencies, dtype=np.float32))
        handle, uf, up = C.c_void_p(), C.c_size_t(), C.c_size_t()
        f, p = self._f[self._pf:], self._p[self._pp:]
        st = _lib.lib().p7x_oprofile_read_pressed(f.ctypes.data, f.shape[0], p.ctypes.data, p.shape[0], bg.ctypes.data,
                                                  C.byref(handle), C.byref(uf), C.byref(up), None)
        if st != 0:
            raise ValueError(f"{self.name}: {_lib.last_error()}") if st == 7 else \
                status_to_exception(st, "p7x_oprofile_read_pressed", _lib.last_error())
        self._pf += uf.value
        self._pp += up.value
        om = OptimizedProfile._from_handle(alphabet, handle)
        om._fill_names()
        return om

    read = __next__


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


# --------------------------------------------------------------------------- results

def _s(b: Optional[bytes]) -> Optional[str]:
    return None if b is None else b.decode()


class Alignment:
    """An alignment of a target domain to the query (reference ``plan7.pyx:229-426``; ``P7_ALIDISPLAY``)."""

    def __init__(self, domain: "Domain"):
        self.domain = domain
        r = domain._rec
        self.hmm_from, self.hmm_to, self.hmm_length = r.hmmfrom, r.hmmto, r.M
        self.hmm_name, self.hmm_accession = _s(r.hmmname), _s(r.hmmacc)
        self.hmm_sequence = _s(r.model)
        self.identity_sequence = _s(r.mline)
        self.target_from, self.target_to, self.target_length = r.sqfrom, r.sqto, r.L
        self.target_name = _s(r.sqname)
        self.target_sequence = _s(r.aseq)
        self.posterior_probabilities = _s(r.ppline)

    def __len__(self) -> int:
        return self.domain._rec.N

    def __str__(self) -> str:
        """One block, as ``p7_alidisplay_Print(fp, ad, 0, -1, FALSE)`` lays it out (reference ``plan7.pyx:262-281``):
        optional RF / MM / CS annotation lines, model line, match line, target line, posterior line."""
        r = self.domain._rec
        hmm, sq = self.hmm_name or "", self.target_name or ""
        nw = max(len(hmm), len(sq))
        cw = max(len(str(v)) for v in (self.hmm_from, self.hmm_to, self.target_from, self.target_to))
        pad = " " * (nw + cw + 1)
        lines = []
        for text, tag in ((_s(r.rfline), "RF"), (_s(r.mmline), "MM"), (_s(r.csline), "CS")):
            if text:
                lines.append(f"  {pad} {text} {tag}")
        lines.append(f"  {hmm:>{nw}} {self.hmm_from:>{cw}d} {self.hmm_sequence} {self.hmm_to:<{cw}d}")
        lines.append(f"  {pad} {self.identity_sequence}")
        lines.append(f"  {sq:>{nw}} {self.target_from:>{cw}d} {self.target_sequence} {self.target_to:<{cw}d}")
        if self.posterior_probabilities:
            lines.append(f"  {pad} {self.posterior_probabilities} PP")
        return "\n".join(lines) + "\n"


class Domain:
    """A single domain of a hit (reference ``plan7.pyx:1441-1688``; ``P7_DOMAIN`` in ``p7_domain.pxd:10-26``)."""

    def __init__(self, hit: "Hit", index: int):
        self.hit = hit
        self._index = index
        self._rec = _lib.DomainRec()
        st = _lib.lib().p7x_tophits_get_domain(hit.hits._handle, hit._index, index, C.byref(self._rec))
        if st != 0:
            raise IndexError("domain index out of range")
        self.alignment = Alignment(self)

    @property
    def env_from(self) -> int:
        return self._rec.ienv

    @property
    def env_to(self) -> int:
        return self._rec.jenv

    @property
    def score(self) -> float:
        return self._rec.bitscore

    @property
    def bias(self) -> float:
        return self._rec.dombias / math.log(2.0)           # plan7.pyx:1536-1554: nats / ln 2

    @property
    def correction(self) -> float:
        return self._rec.domcorrection / math.log(2.0)

    @property
    def envelope_score(self) -> float:
        return self._rec.envsc / math.log(2.0)

    @property
    def pvalue(self) -> float:
        return math.exp(self._rec.lnP)

    @property
    def c_evalue(self) -> float:
        if self.hit.hits.long_targets:
            return math.exp(self._rec.lnP)
        return math.exp(self._rec.lnP) * self.hit.hits.domZ      # plan7.pyx:1557-1574

    @property
    def i_evalue(self) -> float:
        if self.hit.hits.long_targets:
            return math.exp(self._rec.lnP)
        return math.exp(self._rec.lnP) * self.hit.hits.Z

    @property
    def strand(self) -> Optional[str]:
        """"+" / "-" for hits of a `LongTargetsPipeline`, else `None` (reference ``plan7.pyx:1511-1526``)."""
        if not self.hit.hits.long_targets:
            return None
        return "+" if self._rec.iali < self._rec.jali else "-"

    @property
    def reported(self) -> bool:
        return bool(self._rec.is_reported)

    @property
    def included(self) -> bool:
        return bool(self._rec.is_included)

    @property
    def accuracy(self) -> float:
        return self._rec.oasc / (1.0 + abs(self._rec.jenv - self._rec.ienv))


class Domains:
    """Read-only view over the domains of a hit (reference ``plan7.pyx:1850-1935``)."""

    def __init__(self, hit: "Hit"):
        self.hit = hit

    def __len__(self) -> int:
        return self.hit._rec.ndom

    def __getitem__(self, index: int) -> Domain:
        n = len(self)
        if index < 0:
            index += n
        if index < 0 or index >= n:
            raise IndexError("list index out of range")
        return Domain(self.hit, index)

    def __iter__(self):
        return (Domain(self.hit, i) for i in range(len(self)))

    @property
    def reported(self):
        return [d for d in self if d.reported]

    @property
    def included(self):
        return [d for d in self if d.included]


class Hit:
    """A high-scoring target (reference ``plan7.pyx:1936-2235``; ``P7_HIT`` in ``p7_hit.pxd:27-58``)."""

    def __init__(self, hits: "TopHits", index: int):
        self.hits = hits
        self._index = index
        self._rec = _lib.HitRec()
        st = _lib.lib().p7x_tophits_get_hit(hits._handle, index, C.byref(self._rec))
        if st != 0:
            raise IndexError("list index out of range")

    # name / accession / description are read through a fresh record: a setter on another `Hit` object of the same
    # hit replaces the strings (reference ``plan7.pyx:1960-2050``; tests/test_plan7/test_hit.py:29-93)
    def _fresh(self) -> "_lib.HitRec":
        rec = _lib.HitRec()
        if _lib.lib().p7x_tophits_get_hit(self.hits._handle, self._index, C.byref(rec)) != 0:
            raise IndexError("list index out of range")
        return rec

    def _set_text(self, which: int, value: Optional[str]) -> None:
        st = _lib.lib().p7x_tophits_set_hit_text(self.hits._handle, self._index, which, None if value is None else value.encode())
        if st != 0:
            raise ValueError("cannot set hit text")
        self._rec = self._fresh()

    @property
    def name(self) -> str:
        return self._fresh().name.decode()

    @name.setter
    def name(self, name: str) -> None:
        if not isinstance(name, str):
            raise TypeError(f"expected str, found {type(name).__name__}")
        self._set_text(1, name)

    @property
    def seqidx(self) -> int:
        """Index of the target in the searched block / database."""
        return self._rec.seqidx

    @property
    def accession(self) -> Optional[str]:
        return _s(self._fresh().acc)

    @accession.setter
    def accession(self, accession: Optional[str]) -> None:
        self._set_text(2, accession)

    @property
    def description(self) -> Optional[str]:
        return _s(self._fresh().desc)

    @description.setter
    def description(self, description: Optional[str]) -> None:
        self._set_text(4, description)

    @property
    def score(self) -> float:
        return self._rec.score

    @property
    def pre_score(self) -> float:
        return self._rec.pre_score

    @property
    def sum_score(self) -> float:
        return self._rec.sum_score

    @property
    def bias(self) -> float:
        return self._rec.pre_score - self._rec.score             # plan7.pyx:2080-2084

    @property
    def pvalue(self) -> float:
        return math.exp(self._rec.lnP)

    @property
    def evalue(self) -> float:
        if self.hits.long_targets:
            return math.exp(self._rec.lnP)                         # the database size is part of a long-target P-value
        return math.exp(self._rec.lnP) * self.hits.Z              # plan7.pyx:2104-2111

    @property
    def domains(self) -> Domains:
        return Domains(self)

    @property
    def best_domain(self) -> Domain:
        return Domain(self, self._rec.best_domain)

    # flag word: P7X_IS_INCLUDED = 1, P7X_IS_REPORTED = 2, P7X_IS_NEW = 4, P7X_IS_DROPPED = 8, P7X_IS_DUPLICATE = 16;
    # the setters restate plan7.pyx:2125-2235 (the reported / included counts are derived from the flags here)
    def _set_flags(self, flags: int) -> None:
        st = _lib.lib().p7x_tophits_set_hit_flags(self.hits._handle, self._index, flags)
        if st != 0:
            raise IndexError("hit index out of range")
        self._rec.flags = flags

    @property
    def reported(self) -> bool:
        return bool(self._rec.flags & 2)

    @reported.setter
    def reported(self, reported: bool) -> None:
        self._set_flags((self._rec.flags | 2) if reported else (self._rec.flags & ~3))

    @property
    def included(self) -> bool:
        return bool(self._rec.flags & 1)

    @included.setter
    def included(self, included: bool) -> None:
        self._set_flags(((self._rec.flags | 3) & ~24) if included else (self._rec.flags & ~1))

    @property
    def new(self) -> bool:
        return bool(self._rec.flags & 4)

    @new.setter
    def new(self, new: bool) -> None:
        self._set_flags((self._rec.flags | 4) if new else (self._rec.flags & ~4))

    @property
    def dropped(self) -> bool:
        return bool(self._rec.flags & 8)

    @dropped.setter
    def dropped(self, dropped: bool) -> None:
        self._set_flags(((self._rec.flags | 8) & ~1) if dropped else (self._rec.flags & ~8))

    @property
    def duplicate(self) -> bool:
        return bool(self._rec.flags & 16)

    @duplicate.setter
    def duplicate(self, duplicate: bool) -> None:
        self._set_flags(((self._rec.flags | 16) & ~3) if duplicate else (self._rec.flags & ~16))

    @property
    def length(self) -> int:
        return Domain(self, 0)._rec.L

    # domain number estimation columns of the tabular output (p7_hit.pxd:42-51)
    @property
    def nexpected(self) -> float:
        return self._rec.nexpected

    @property
    def nregions(self) -> int:
        return self._rec.nregions

    @property
    def nclustered(self) -> int:
        return self._rec.nclustered

    @property
    def noverlaps(self) -> int:
        return self._rec.noverlaps

    @property
    def nenvelopes(self) -> int:
        return self._rec.nenvelopes


class HitHandles:
    """The results of a batch of queries as bare library handles (``p7x_tophits *``), owned by this object: what
    ``Pipeline._search_finish_batch(raw=True)`` returns.  ``array`` goes to the C entry points that take ``p7x_tophits **``;
    the handles are destroyed together when the object goes."""

    __slots__ = ("array", "n")

    def __init__(self, array, n: int):
        self.array = array
        self.n = int(n)

    def __len__(self) -> int:
        return self.n

    def __del__(self):
        array, self.array = getattr(self, "array", None), None
        if array is not None:
            try:
                _lib.lib().p7x_tophits_destroy_many(array, self.n)
            except Exception:
                pass


class TopHits:
    """An ordered list of hits with the pipeline accounting that produced it
    (reference ``plan7.pyx:8312-9276``; ``P7_TOPHITS`` + copied ``P7_PIPELINE``)."""

    def __init__(self, query=None, _handle=None):
        self.query = query
        self._handle = _handle

    def __del__(self):
        if getattr(self, "_handle", None):
            try:
                _lib.lib().p7x_tophits_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    def _cfg(self) -> "_lib.PipelineCfg":
        cfg = _lib.PipelineCfg()
        _lib.lib().p7x_tophits_get_cfg(self._handle, C.byref(cfg))
        return cfg

    def _ctr(self) -> "_lib.Counters":
        c = _lib.Counters()
        _lib.lib().p7x_tophits_get_counters(self._handle, C.byref(c))
        return c

    def __len__(self) -> int:
        return int(_lib.lib().p7x_tophits_nhits(self._handle))

    def __bool__(self) -> bool:
        return len(self) > 0

    def __getitem__(self, index: int) -> Hit:
        n = len(self)
        if index < 0:
            index += n
        if index < 0 or index >= n:
            raise IndexError("list index out of range")
        return Hit(self, index)

    def __iter__(self):
        return (Hit(self, i) for i in range(len(self)))

    @property
    def Z(self) -> float:
        return self._cfg().Z

    @property
    def domZ(self) -> float:
        return self._cfg().domZ

    @property
    def E(self) -> float:
        return self._cfg().E

    # the thresholds the hits were reported / included with (reference plan7.pyx:8600-8690): a bit-score threshold is None
    # while its E-value twin is the one in force
    T = property(lambda self: None if self._cfg().by_E else self._cfg().T)
    domE = property(lambda self: self._cfg().domE)
    domT = property(lambda self: None if self._cfg().dom_by_E else self._cfg().domT)
    incE = property(lambda self: self._cfg().incE)
    incT = property(lambda self: None if self._cfg().inc_by_E else self._cfg().incT)
    incdomE = property(lambda self: self._cfg().incdomE)
    incdomT = property(lambda self: None if self._cfg().incdom_by_E else self._cfg().incdomT)
    bit_cutoffs = property(lambda self: {1: "gathering", 2: "noise", 3: "trusted"}.get(int(self._cfg().use_bit_cutoffs)))

    @property
    def searched_models(self) -> int:
        return self._ctr().nmodels

    @property
    def searched_nodes(self) -> int:
        return self._ctr().nnodes

    @property
    def searched_sequences(self) -> int:
        return self._ctr().nseqs

    @property
    def searched_residues(self) -> int:
        return self._ctr().nres

    @property
    def stage_counts(self) -> dict:
        """n_past_{msv,bias,vit,fwd} (``p7_pipeline.pxd:88-101``; pickled by the reference at ``plan7.pyx:8457-8460``)."""
        c = self._ctr()
        return dict(msv=c.n_past_msv, bias=c.n_past_bias, vit=c.n_past_vit, fwd=c.n_past_fwd)

    @property
    def guard_counts(self) -> dict:
        """How often the two guards acted in this search: targets the F3 guard took back out of the device's survivor
        list, device envelopes the optimal-accuracy near-tie guard had the host twin repeat."""
        a, b, why = C.c_int64(0), C.c_int64(0), (C.c_int64 * 8)()
        _lib.lib().p7x_tophits_get_guard_counts(self._handle, C.byref(a), C.byref(b), why)
        e0, e1, e2 = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.lib().p7x_tophits_get_ensemble_counts(self._handle, C.byref(e0), C.byref(e1), C.byref(e2))
        return {"f3_dropped": int(a.value), "oa_redone": int(b.value), "ens_device": int(e0.value), "ens_redone": int(e1.value),
                "region_redone": int(e2.value),
                "oa_why": dict(zip(("match", "insert", "delete", "c_from_e", "j_from_e", "end_cell", "begin", "pp_digit"), map(int, why)))}

    @property
    def timings_ms(self) -> dict:
        """Milliseconds of the batch this query travelled in.  ``host_domaindef``: the host stage, wall; of that
        ``ensemble_wait`` and ``envelope_wait`` were spent waiting for the device (ensemble kernels; the two envelope rounds)
        and ``host_stage_busy`` is the rest -- what the host itself did (region bookkeeping, clustering, alignments, hit
        lists).  ``host_multi``: the host's own share of the multi-domain regions (clustering; sampling too when the ensembles
        stay on the host).  ``envelopes``: wall from the first launch of the stage to the last result."""
        buf = (C.c_double * 20)()
        _lib.lib().p7x_tophits_get_timings(self._handle, buf, 20)
        return dict(zip(("msv", "bias", "viterbi", "forward", "fwd_rows", "host_domaindef", "total", "msv_kernel",
                         "envelopes", "host_multi", "stage1", "stage2", "host_stage_busy", "ensemble_wait", "envelope_wait",
                         # msv_kernel is the batch's largest fast-MSV launch by its own HIP events; what it covered:
                         "batch_queries", "msv_launch_lanes", "msv_launch_nodes"), buf))

    @property
    def reported(self):
        return [h for h in self if h.reported]

    @property
    def included(self):
        return [h for h in self if h.included]

    @property
    def mode(self) -> str:
        """``"search"`` or ``"scan"`` (reference ``plan7.pyx:8560-8571``)."""
        return "scan" if self._cfg().mode == _P7X_SCAN_MODELS else "search"

    @property
    def long_targets(self) -> bool:
        """Whether these hits come from a `LongTargetsPipeline` (reference ``plan7.pyx:8739-8746``)."""
        return bool(self._cfg().long_targets)

    @property
    def strand(self) -> Optional[str]:
        """The strand a `LongTargetsPipeline` was restricted to, or `None` (reference ``plan7.pyx:8748-8764``)."""
        c = self._cfg()
        if c.long_targets:
            return {1: "watson", 2: "crick"}.get(c.strands)
        return None

    @property
    def block_length(self) -> Optional[int]:
        c = self._cfg()
        return int(c.block_length) if c.long_targets else None

    def sort(self, by: str = "key") -> None:
        """``p7_tophits_SortBySortkey`` / ``p7_tophits_SortBySeqidxAndAlipos`` (reference ``plan7.pyx:9120-9148``)."""
        if by == "key":
            _lib.lib().p7x_tophits_sort_by_key(self._handle)
        elif by == "seqidx":
            _lib.lib().p7x_tophits_sort_by_seqidx(self._handle)
        else:
            raise InvalidParameter("by", by, choices=["key", "seqidx"])

    def is_sorted(self, by: str = "key") -> bool:
        if by not in ("key", "seqidx"):
            raise InvalidParameter("by", by, choices=["key", "seqidx"])
        return bool(_lib.lib().p7x_tophits_is_sorted(self._handle, 1 if by == "seqidx" else 0))

    def threshold(self) -> None:
        """Re-apply the reporting / inclusion thresholds (``p7_tophits_Threshold``, ``plan7.pyx:8804-8818``)."""
        _lib.lib().p7x_tophits_threshold(self._handle)

    def write(self, fh, format: str = "targets", header: bool = True) -> None:
        """Tabular output as ``hmmsearch --tblout`` (``"targets"``) / ``--domtblout`` (``"domains"``) write it:
        ``p7_tophits_TabularTargets`` / ``p7_tophits_TabularDomains`` behind reference ``plan7.pyx:9071-9118``, pinned by
        the golden ``.tbl`` / ``.domtbl`` files.  ``"pfam"`` (``plan7.pyx:9155-9164``) is ``p7_tophits_TabularXfam``: the
        reported hits, a blank line, then every reported domain as a row of its own, best first (by bit score, or by E-value if
        inclusion is by E-value); it ignores ``header``.  No file in the reference pins that third layout, so it is restated from
        upstream's format strings and NOT verified byte for byte.  ``fh`` is a file object opened in binary mode."""
        if format not in ("targets", "domains", "pfam"):
            raise InvalidParameter("format", format, choices=["targets", "domains", "pfam"])
        if format == "pfam":
            fh.write(self._xfam().encode())
            return
        q = self.query
        qname = getattr(q, "name", None) if q is not None and not isinstance(q, str) else q
        qacc = getattr(q, "accession", None) if q is not None and not isinstance(q, str) else None
        qname = qname if qname else "-"
        qacc = qacc if qacc else "-"
        hits = [h for h in self if h.reported]
        tnamew = max([20] + [len(h.name) for h in self])
        taccw = max([10] + [len(h.accession or "") for h in self])
        qnamew = max(20, len(qname))
        qaccw = max(10, len(qacc))
        Z, domZ = self.Z, self.domZ
        out = []
        if format == "targets":
            if header:
                out.append("#%*s %22s %22s %33s" % (tnamew + qnamew + taccw + qaccw + 2, "", "--- full sequence ----",
                                                     "--- best 1 domain ----", "--- domain number estimation ----"))
                out.append("#%-*s %-*s %-*s %-*s %9s %6s %5s %9s %6s %5s %5s %3s %3s %3s %3s %3s %3s %3s %s" % (
                    tnamew - 1, " target name", taccw, "accession", qnamew, "query name", qaccw, "accession",
                    "  E-value", " score", " bias", "  E-value", " score", " bias", "exp", "reg", "clu", " ov", "env",
                    "dom", "rep", "inc", "description of target"))
                out.append("#%*s %*s %*s %*s %9s %6s %5s %9s %6s %5s %5s %3s %3s %3s %3s %3s %3s %3s %s" % (
                    tnamew - 1, "-------------------", taccw, "----------", qnamew, "--------------------", qaccw,
                    "----------", "---------", "------", "-----", "---------", "------", "-----", "---", "---", "---",
                    "---", "---", "---", "---", "---", "---------------------"))
            for h in hits:
                r, d = h._rec, h.best_domain._rec
                out.append("%-*s %-*s %-*s %-*s %9.2g %6.1f %5.1f %9.2g %6.1f %5.1f %5.1f %3d %3d %3d %3d %3d %3d %3d %s" % (
                    tnamew, h.name, taccw, h.accession or "-", qnamew, qname, qaccw, qacc,
                    math.exp(r.lnP) * Z, r.score, r.pre_score - r.score, math.exp(d.lnP) * Z, d.bitscore,
                    d.dombias / math.log(2.0), r.nexpected, r.nregions, r.nclustered, r.noverlaps, r.nenvelopes, r.ndom,
                    r.nreported, r.nincluded, h.description or "-"))
        else:
            qlen = getattr(q, "M", None)
            if qlen is None:
                qlen = len(q) if q is not None and not isinstance(q, str) else 0
            if header:
                out.append("#%*s %22s %40s %11s %11s %11s" % (tnamew + qnamew - 1 + 15 + taccw + qaccw, "",
                                                               "--- full sequence ---", "-------------- this domain -------------",
                                                               "hmm coord", "ali coord", "env coord"))
                out.append("#%-*s %-*s %5s %-*s %-*s %5s %9s %6s %5s %3s %3s %9s %9s %6s %5s %5s %5s %5s %5s %5s %5s %4s %s" % (
                    tnamew - 1, " target name", taccw, "accession", "tlen", qnamew, "query name", qaccw, "accession", "qlen",
                    "E-value", "score", "bias", "#", "of", "c-Evalue", "i-Evalue", "score", "bias", "from", "to", "from",
                    "to", "from", "to", "acc", "description of target"))
                out.append("#%*s %*s %5s %*s %*s %5s %9s %6s %5s %3s %3s %9s %9s %6s %5s %5s %5s %5s %5s %5s %5s %4s %s" % (
                    tnamew - 1, "-------------------", taccw, "----------", "-----", qnamew, "--------------------", qaccw,
                    "----------", "-----", "---------", "------", "-----", "---", "---", "---------", "---------", "------",
                    "-----", "-----", "-----", "-----", "-----", "-----", "-----", "----", "---------------------"))
            for h in hits:
                r = h._rec
                nd = 0
                for dom in h.domains:
                    d = dom._rec
                    if not d.is_reported:
                        continue
                    nd += 1
                    out.append("%-*s %-*s %5d %-*s %-*s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5d %5d %5d %5d %4.2f %s" % (
                        tnamew, h.name, taccw, h.accession or "-", d.L, qnamew, qname, qaccw, qacc, qlen,
                        math.exp(r.lnP) * Z, r.score, r.pre_score - r.score, nd, r.nreported, math.exp(d.lnP) * domZ,
                        math.exp(d.lnP) * Z, d.bitscore, d.dombias / math.log(2.0), d.hmmfrom, d.hmmto, d.sqfrom, d.sqto,
                        d.ienv, d.jenv, d.oasc / (1.0 + abs(float(d.jenv - d.ienv))), h.description or "-"))
        fh.write(("\n".join(out) + "\n").encode() if out else b"")

    def _xfam(self) -> str:
        q = self.query
        qname = (getattr(q, "name", None) if q is not None and not isinstance(q, str) else q) or "-"
        hits = [h for h in self if h.reported]
        tnamew = max([20] + [len(h.name) for h in self])
        taccw = max([10] + [len(h.accession or "") for h in self])
        qnamew = max(20, len(qname))
        Z = self.Z
        ln2 = math.log(2.0)
        o = []
        if self.long_targets:
            posw = max([7] + [len(str(max(d._rec.iali, d._rec.jali, d._rec.ienv, d._rec.jenv, d._rec.L))) for h in self for d in h.domains])
            o.append("# hit scores\n# ----------\n#\n")
            o.append("# %-*s %-*s %-*s %6s %9s %5s  %s  %s %6s %*s %*s %*s %*s %*s   %s\n" % (
                tnamew - 1, "name", taccw, "acc", qnamew, "query", "bits", "  e-value", " bias", "hmm-st", "hmm-en", "strand",
                posw, "ali-st", posw, "ali-en", posw, "env-st", posw, "env-en", posw, "sq-len", "description of target"))
            o.append("# %*s %*s %*s %6s %9s %5s %s %s %6s %*s %*s %*s %*s %*s   %s\n" % (
                tnamew - 1, "-------------------", taccw, "----------", qnamew, "--------------------", "------", "---------", "-----",
                "-------", "-------", "------", posw, "-------", posw, "-------", posw, "-------", posw, "-------", posw, "-------",
                "---------------------"))
            for h in hits:
                r, d = h._rec, h.best_domain._rec
                o.append("%-*s  %-*s %-*s %6.1f %9.2g %5.1f %7d %7d %s %*d %*d %*d %*d %*d   %s\n" % (
                    tnamew, h.name, taccw, h.accession or "-", qnamew, qname, r.score, math.exp(r.lnP), d.dombias / ln2, d.hmmfrom, d.hmmto,
                    "   +  " if d.iali < d.jali else "   -  ", posw, d.iali, posw, d.jali, posw, d.ienv, posw, d.jenv, posw, d.L,
                    h.description or "-"))
            return "".join(o)
        o.append("# Sequence scores\n# ---------------\n#\n")
        o.append("# %-*s %6s %9s %3s %5s %5s    %s\n" % (tnamew - 1, "name", " bits", "  E-value", "n", "exp", " bias", "description"))
        o.append("# %*s %6s %9s %3s %5s %5s    %s\n" % (tnamew - 1, "-------------------", "------", "---------", "---", "-----", "-----",
                                                        "---------------------"))
        rows = []
        for h in hits:
            r = h._rec
            o.append("%-*s  %6.1f %9.2g %3d %5.1f %5.1f    %s\n" % (tnamew, h.name, r.score, math.exp(r.lnP) * Z, r.ndom, r.nexpected, 0.0,
                                                                  h.description or "-"))
            nrep = 0
            for dom in h.domains:
                if dom._rec.is_reported:
                    nrep += 1
                    rows.append((h, dom._rec, nrep))
        o.append("\n")
        by_E = bool(self._cfg().inc_by_E)
        order = sorted(range(len(rows)), key=lambda i: (-(-rows[i][1].lnP if by_E else rows[i][1].bitscore), rows[i][0].name, i))
        o.append("# Domain scores\n# -------------\n#\n")
        o.append("# %-*s %6s %9s %5s %5s %6s %6s %6s %6s %6s %6s     %s\n" % (tnamew - 1, " name", "bits", "E-value", "hit", "bias", "env-st", "env-en",
                                                                            "ali-st", "ali-en", "hmm-st", "hmm-en", "description"))
        o.append("# %*s %6s %9s %5s %5s %6s %6s %6s %6s %6s %6s      %s\n" % (tnamew - 1, "-------------------", "------", "---------", "-----", "-----",
                                                                           "------", "------", "------", "------", "------", "------",
                                                                           "---------------------"))
        for i in order:
            h, d, nth = rows[i]
            o.append("%-*s  %6.1f %9.2g %5d %5.1f %6d %6d %6d %6d %6d %6d     %s\n" % (
                tnamew, h.name, d.bitscore, math.exp(d.lnP) * Z, nth, d.dombias / ln2, d.ienv, d.jenv, d.sqfrom, d.sqto, d.hmmfrom, d.hmmto,
                h.description or "-"))
        return "".join(o)

    def copy(self) -> "TopHits":
        h = _lib.lib().p7x_tophits_clone(self._handle)
        return TopHits(self.query, C.c_void_p(h))

    def to_bytes(self) -> bytes:
        """Flat byte image (``p7x_tophits_serialize``) for shipping results between processes / ranks."""
        n = _lib.lib().p7x_tophits_serialize(self._handle, None, 0)
        buf = C.create_string_buffer(int(n))
        _lib.lib().p7x_tophits_serialize(self._handle, buf, n)
        return buf.raw

    @classmethod
    def from_bytes(cls, data: bytes, query=None) -> "TopHits":
        h = _lib.lib().p7x_tophits_deserialize(data, len(data))
        if not h:
            raise ValueError(_lib.last_error() or "cannot deserialise TopHits")
        return cls(query, C.c_void_p(h))

    def __reduce__(self):
        return (TopHits.from_bytes, (self.to_bytes(), self.query if isinstance(self.query, str) else None))

    def merge(self, *others: "TopHits") -> "TopHits":
        """Reference ``plan7.pyx:9172-9276``: concatenate, sum the accounting, re-threshold with the global Z."""
        merged = self.copy()
        for o in others:
            st = _lib.lib().p7x_tophits_merge(merged._handle, o._handle)
            if st != 0:
                raise ValueError(_lib.last_error())
        return merged

    def __add__(self, other: "TopHits") -> "TopHits":
        return self.merge(other)

    @staticmethod
    def merge_many(blobs, queries=None, threads: int = 0) -> List["TopHits"]:
        """The merging side of a sharded many-query search: ``blobs[r][q]`` is ``to_bytes()`` of query ``q`` on shard
        ``r`` (``None`` / ``b""`` for a shard with nothing to report).  One native call (``p7x_tophits_merge_many``),
        threaded over the queries; the result for query ``q`` equals ``shard0[q].merge(shard1[q], ...)``."""
        nparts = len(blobs)
        nq = len(blobs[0]) if nparts else 0
        if any(len(b) != nq for b in blobs):
            raise ValueError("every shard must report the same number of queries")
        if nq == 0:
            return []
        flat = (C.c_char_p * (nq * nparts))()
        sizes = (C.c_size_t * (nq * nparts))()
        for r, shard in enumerate(blobs):
            for q, b in enumerate(shard):
                flat[q * nparts + r] = b if b else None
                sizes[q * nparts + r] = len(b) if b else 0
        outs = (C.c_void_p * nq)()
        st = _lib.lib().p7x_tophits_merge_many(flat, sizes, nq, nparts, int(threads), outs)
        if st != 0:
            raise ValueError(_lib.last_error())
        return [TopHits(None if queries is None else queries[q], C.c_void_p(outs[q])) for q in range(nq)]


# --------------------------------------------------------------------------- Pipeline

class Pipeline:
    """An accelerated sequence/profile comparison pipeline (reference ``plan7.pyx:5423-6906``).

    Same constructor keywords and defaults as the reference (``plan7.pyx:5413-5421``).  ``search_hmm`` runs
    ``Pipeline._search_loop`` (``plan7.pyx:6393-6453``) for a whole ``DigitalSequenceBlock`` on one MI355X.
    """

    M_HINT = 100
    L_HINT = 100
    _BIT_CUTOFFS = {"gathering": 1, "noise": 2, "trusted": 3}

    def __init__(self, alphabet: Alphabet, background: Optional[Background] = None, *, bias_filter: bool = True,
                 null2: bool = True, seed: int = 42, Z=None, domZ=None, F1: float = 0.02, F2: float = 1e-3,
                 F3: float = 1e-5, E: float = 10.0, T=None, domE: float = 10.0, domT=None, incE: float = 0.01,
                 incT=None, incdomE: float = 0.01, incdomT=None, bit_cutoffs: Optional[str] = None,
                 device: int = 0, host_threads: int = 0, host_envelopes: bool = False, host_regions: bool = False,
                 oa_guard: Optional[float] = None, host_ensembles: bool = False, ens_guard: Optional[float] = None):
        self.alphabet = alphabet
        if background is None:
            self.background = Background(alphabet)
        elif background.alphabet != alphabet:
            raise AlphabetMismatch(alphabet, background.alphabet)
        else:
            self.background = background.copy()
        for name, v in (("F1", F1), ("F2", F2), ("F3", F3)):
            if not (0.0 <= v <= 1.0) and v != 1.0:
                raise InvalidParameter(name, v, hint="real number between 0 and 1")
        for name, v in (("E", E), ("domE", domE), ("incE", incE), ("incdomE", incdomE)):
            if v < 0:
                raise InvalidParameter(name, v, hint="positive real number")
        if bit_cutoffs is not None and bit_cutoffs not in self._BIT_CUTOFFS:
            raise InvalidParameter("bit_cutoffs", bit_cutoffs, choices=list(self._BIT_CUTOFFS) + [None])
        self.bias_filter, self.null2, self.seed = bias_filter, null2, seed
        self.Z, self.domZ = Z, domZ
        self.F1, self.F2, self.F3 = F1, F2, F3
        self.E, self.T, self.domE, self.domT = E, T, domE, domT
        self.incE, self.incT, self.incdomE, self.incdomT = incE, incT, incdomE, incdomT
        self.bit_cutoffs = bit_cutoffs
        self.device = device
        self.host_threads = host_threads
        self.host_envelopes = int(host_envelopes)       # bool for the protein pipeline; long targets: 0 auto, 1 host, 2 device
        self.host_regions = bool(host_regions)
        self.host_ensembles = bool(host_ensembles)      # True: the stochastic traceback ensembles stay on the host workers
        self.oa_guard = oa_guard          # None: the library's default (p7x_pipeline_cfg.oa_guard)
        self.ens_guard = ens_guard        # None: the library's default (p7x_pipeline_cfg.ens_guard)
        self._mode = _P7X_SEARCH_SEQS
        self._db_cache = None           # (id(block), block version, n, device) -> SequenceDatabase

    def clear(self) -> None:
        """Reference ``plan7.pyx:6113-6154``: reset accounting between queries (stateless here)."""

    def _cfg(self) -> "_lib.PipelineCfg":
        c = _lib.PipelineCfg()
        _lib.lib().p7x_pipeline_cfg_default(C.byref(c))
        c.do_biasfilter, c.do_null2, c.seed = int(self.bias_filter), int(self.null2), int(self.seed or 0)
        c.F1, c.F2, c.F3 = self.F1, self.F2, self.F3
        c.E, c.domE, c.incE, c.incdomE = self.E, self.domE, self.incE, self.incdomE
        if self.T is not None:
            c.T, c.by_E = float(self.T), 0
        if self.domT is not None:
            c.domT, c.dom_by_E = float(self.domT), 0
        if self.incT is not None:
            c.incT, c.inc_by_E = float(self.incT), 0
        if self.incdomT is not None:
            c.incdomT, c.incdom_by_E = float(self.incdomT), 0
        if self.Z is not None:
            c.Z, c.Z_setby = float(self.Z), 1
        if self.domZ is not None:
            c.domZ, c.domZ_setby = float(self.domZ), 1
        c.use_bit_cutoffs = 0 if self.bit_cutoffs is None else self._BIT_CUTOFFS[self.bit_cutoffs]
        c.host_threads = int(self.host_threads)
        c.host_envelopes = int(self.host_envelopes)      # long targets also take 2: always the device (see p7x.h)
        c.host_regions = int(self.host_regions)
        c.host_ensembles = int(self.host_ensembles)
        if self.oa_guard is not None:
            c.oa_guard = float(self.oa_guard)
        if getattr(self, "ens_guard", None) is not None:
            c.ens_guard = float(self.ens_guard)
        c.mode = int(self._mode)
        return c

    def _get_om_from_query(self, query, L: int = L_HINT) -> OptimizedProfile:
        """Reference ``plan7.pyx:5979-6013``."""
        if isinstance(query, OptimizedProfile):
            return query
        if isinstance(query, Profile):
            return query.to_optimized()
        if isinstance(query, HMM):
            return OptimizedProfile(query, self.background, L)
        raise TypeError(f"Expected HMM, Profile or OptimizedProfile, found {type(query).__name__}")

    def search_hmm(self, query, sequences, database: Optional["SequenceDatabase"] = None) -> TopHits:
        """Run the pipeline with ``query`` against every target of ``sequences``
        (reference ``plan7.pyx:6156-6262``)."""
        if query.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, query.alphabet)
        if isinstance(sequences, SequenceFile):
            if not sequences.digital:
                raise ValueError("target sequences file is not in digital mode")
            sequences = sequences.read_block()
        if isinstance(sequences, SequenceDatabase):
            database, sequences = sequences, sequences.block
        if database is not None and sequences is None:        # database built by SequenceDatabase.from_packed
            if database.alphabet != self.alphabet:
                raise AlphabetMismatch(self.alphabet, database.alphabet)
            return self._search_database(query, database)
        if not isinstance(sequences, DigitalSequenceBlock):
            raise TypeError(f"Expected DigitalSequenceBlock or SequenceFile, found {type(sequences).__name__}")
        if sequences.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, sequences.alphabet)
        L = len(sequences[0]) if len(sequences) else self.L_HINT
        om = self._get_om_from_query(query, L)
        if database is None:
            key = (id(sequences), getattr(sequences, "_version", 0), len(sequences), self.device)
            if self._db_cache is None or self._db_cache[0] != key:
                self._db_cache = (key, SequenceDatabase(sequences, device=self.device))
            database = self._db_cache[1]
        return self._search_database(om, database, query)

    def scan_seq(self, query: DigitalSequence, optimized_profiles) -> TopHits:
        """Run the pipeline with one query sequence against a collection of profiles (reference
        ``plan7.pyx:6534-6622``).  See :func:`pyhmmer_amd.hmmer.hmmscan` for the batched form."""
        from .hmmer import hmmscan
        if query.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, query.alphabet)
        opts = dict(bias_filter=self.bias_filter, null2=self.null2, seed=self.seed, Z=self.Z, domZ=self.domZ, F1=self.F1,
                    F2=self.F2, F3=self.F3, E=self.E, T=self.T, domE=self.domE, domT=self.domT, incE=self.incE,
                    incT=self.incT, incdomE=self.incdomE, incdomT=self.incdomT, bit_cutoffs=self.bit_cutoffs,
                    background=self.background)
        return next(iter(hmmscan(query, optimized_profiles, cpus=self.host_threads, devices=[self.device], **opts)))

    # -- the two stages of a search.  ``hmmer.hmmsearch`` runs stage 1 (device filters + parsers, ``p7x_search_batch_enqueue``
    #    / ``p7x_search_block_wait``) of the next queries while stage 2 (domain definition, ``p7x_search_batch_finish``) of the
    #    previous ones is still busy on the host.  A pending search is the tuple (handle, profiles, database, labels) of a
    #    BATCH of queries that share one set of device launches; a single query is a batch of one.  Both halves of
    #    stage 1 must run on the same thread.
    def _search_enqueue_batch(self, queries, database: "SequenceDatabase", labels=None):
        oms = []
        for q in queries:
            if q.alphabet != self.alphabet:
                raise AlphabetMismatch(self.alphabet, q.alphabet)
            oms.append(self._get_om_from_query(q, self.L_HINT))
        cfg = self._cfg()
        pend = C.c_void_p()
        bgf = np.ascontiguousarray(self.background.residue_frequencies, dtype=np.float32)
        handles = (C.c_void_p * len(oms))(*[om._handle for om in oms])
        st = _lib.lib().p7x_search_batch_enqueue(C.byref(cfg), handles, len(oms), bgf.ctypes.data, database._handle, C.byref(pend))
        if st == 11 and self.bit_cutoffs is not None and "cutoffs" in _lib.last_error():
            missing = next((om for om in oms if not getattr(om.cutoffs, self.bit_cutoffs + "_available")()), oms[0])
            raise MissingCutoffs(missing.name, self.bit_cutoffs)       # plan7.pyx:6424-6425
        if st != 0:
            raise status_to_exception(st, "p7x_search_batch_enqueue", _lib.last_error())
        return (pend, oms, database, list(labels) if labels is not None else list(queries))

    def _search_enqueue(self, query, database: "SequenceDatabase", label=None):
        return self._search_enqueue_batch([query], database, None if label is None else [label])

    def _search_begin(self, query, database: "SequenceDatabase", label=None):
        pending = self._search_enqueue(query, database, label)
        try:
            self._search_wait(pending)
        except BaseException:
            _lib.lib().p7x_pending_destroy(pending[0])
            raise
        return pending

    @staticmethod
    def _search_wait(pending) -> None:
        st = _lib.lib().p7x_search_block_wait(pending[0])
        if st != 0:
            raise status_to_exception(st, "p7x_search_block_wait", _lib.last_error())

    @staticmethod
    def _search_finish_batch(pending, raw: bool = False):
        """Stage 2 of a batch: one ``TopHits`` per query -- or, with ``raw``, the library's handles as they are
        (``HitHandles``): the scan orientation only passes the per-model lists on to ``p7x_scan_accum_add_indexed``, and
        20,000 Python objects per pass, each destroyed through its own foreign call, were a fifth of a resident pass."""
        pend, oms, database, labels = pending
        outs = (C.c_void_p * len(oms))()
        st = _lib.lib().p7x_search_batch_finish(pend, database._names, database._accs, database._descs, outs)
        if st != 0:
            raise status_to_exception(st, "p7x_search_batch_finish", _lib.last_error())
        if raw:
            return HitHandles(outs, len(oms))
        res = []
        for om, label, h in zip(oms, labels, outs):
            hits = TopHits(label, C.c_void_p(h))
            hits._om = om                   # the alignments refer to the profile: keep it alive with the hits
            res.append(hits)
        return res

    @staticmethod
    def _search_finish(pending) -> TopHits:
        return Pipeline._search_finish_batch(pending)[0]

    def _search_database(self, query, database: "SequenceDatabase", label=None) -> TopHits:
        om = self._get_om_from_query(query, self.L_HINT)
        cfg = self._cfg()
        out = C.c_void_p()
        bgf = np.ascontiguousarray(self.background.residue_frequencies, dtype=np.float32)
        st = _lib.lib().p7x_search_block(C.byref(cfg), om._handle, bgf.ctypes.data, database._handle,
                                         database._names, database._accs, database._descs, C.byref(out))
        if st == 11 and self.bit_cutoffs is not None:
            raise MissingCutoffs(om.name, self.bit_cutoffs)       # plan7.pyx:6424-6425
        if st != 0:
            raise status_to_exception(st, "p7x_search_block", _lib.last_error())
        return TopHits(label if label is not None else query, out)


class LongTargetsPipeline(Pipeline):
    """An HMMER3 pipeline tuned for long (nucleotide) targets: ``nhmmer`` (reference ``plan7.pyx:6917-7763``).

    Same constructor as the reference (``plan7.pyx:6957-7060``: ``F1=0.02, F2=3e-3, F3=3e-5, strand, B1=100, B2=240,
    B3=1000, block_length=0x40000, window_length, window_beta``).  ``search_hmm`` runs
    ``_search_loop_longtargets`` + the E-value / duplicate / threshold tail (``plan7.pyx:7272-7418``): the SSV scan of
    every target strand on the device, the windows it seeds on the host (``p7x_search_longtargets``).
    ``window_beta`` (recomputing ``max_length`` from the core model, ``p7_Builder_MaxLength``) belongs to the HMM
    builder, which is out of scope: the model's own ``MAXL`` is used unless ``window_length`` is given."""

    _STRANDS = {None: 0, "watson": 1, "crick": 2}

    def __init__(self, alphabet: Alphabet, background: Optional[Background] = None, *, F1: float = 0.02, F2: float = 3e-3,
                 F3: float = 3e-5, strand: Optional[str] = None, B1: int = 100, B2: int = 240, B3: int = 1000,
                 block_length: int = 0x40000, window_length: Optional[int] = None, window_beta: Optional[float] = None,
                 **kwargs):
        if not (alphabet.is_dna() or alphabet.is_rna()):
            raise ValueError(f"Expected nucleotide alphabet, found {alphabet!r}")          # plan7.pyx:7062-7064
        if strand not in self._STRANDS:
            raise InvalidParameter("strand", strand, choices=["watson", "crick", None])
        for name, v in (("B1", B1), ("B2", B2), ("B3", B3)):
            if v < 0:
                raise InvalidParameter(name, v, hint="positive integer")
        if block_length <= 0:
            raise InvalidParameter("block_length", block_length, hint="strictly positive integer")
        if window_length is not None and window_length < 4:
            raise InvalidParameter("window_length", window_length, hint="integer greater than or equal to 4")
        if window_beta is not None and not (0 < window_beta <= 1):
            raise InvalidParameter("window_beta", window_beta, hint="real number between 0 and 1")
        super().__init__(alphabet, background, F1=F1, F2=F2, F3=F3, **kwargs)
        self.strand = strand
        self.B1, self.B2, self.B3 = int(B1), int(B2), int(B3)
        self.block_length = int(block_length)
        self.window_length = window_length
        self.window_beta = window_beta

    def _cfg(self) -> "_lib.PipelineCfg":
        c = super()._cfg()
        c.long_targets = 1
        c.strands = self._STRANDS[self.strand]
        c.B1, c.B2, c.B3 = self.B1, self.B2, self.B3
        c.block_length = self.block_length
        c.window_length = -1 if self.window_length is None else int(self.window_length)
        return c

    def _windowed_om(self, query, L: int, cfg) -> "OptimizedProfile":
        """The optimized profile of ``query`` and the two window lengths of the search (reference ``plan7.pyx:7336-7354``):
        ``window_length`` overrides everything; otherwise the scan uses the ``max_length`` the profile was built with (the
        file's ``MAXL``) while the E-values use ``p7_Builder_MaxLength(hmm, window_beta)``, which -- as in the reference --
        also replaces ``hmm.max_length`` for later searches.  An HMM without ``MAXL`` gets the computed bound for both."""
        if self.window_length is None and isinstance(query, HMM):
            view, keep = query._view()
            w = C.c_int32(0)
            beta = 1e-7 if self.window_beta is None else float(self.window_beta)     # p7_DEFAULT_WINDOW_BETA, plan7.pyx:6993
            st = _lib.lib().p7x_hmm_max_length(C.byref(view), beta, C.byref(w))
            if st != 0:
                raise status_to_exception(st, "p7x_hmm_max_length", _lib.last_error())
            if (query.max_length or -1) <= 0:
                query.max_length = int(w.value)
            om = self._get_om_from_query(query, L)
            query.max_length = int(w.value)
            cfg.evalue_window_length = int(w.value)
            return om
        return self._get_om_from_query(query, L)

    @staticmethod
    def _pack(sequences):
        """Flat arrays of the C-ABI (255 x1..xL 255 ...) with 64-bit lengths, and the name tables."""
        n = len(sequences)
        lengths = np.array([len(s) for s in sequences], dtype=np.int64)
        offsets = np.empty(n, dtype=np.int64)
        dsq = np.full(int(lengths.sum()) + n + 1, 255, dtype=np.uint8)
        pos = 1
        for i, s in enumerate(sequences):
            offsets[i] = pos
            dsq[pos:pos + len(s)] = s.sequence
            pos += len(s) + 1
        names = (C.c_char_p * max(n, 1))(*[s.name.encode() for s in sequences])
        accs = (C.c_char_p * max(n, 1))(*[(s.accession or "").encode() for s in sequences])
        descs = (C.c_char_p * max(n, 1))(*[(s.description or "").encode() for s in sequences])
        return dsq, offsets, lengths, names, accs, descs

    def _prepare_query(self, query, sequences):
        """The per-query state of a search -- the optimized profile, the configuration record with the two window lengths,
        and the replacement of ``hmm.max_length`` (reference ``plan7.pyx:7336-7354``).  It reads and writes the query object,
        so it runs on the caller's thread, in query order, before a search is handed to a worker: ``hmmer.nhmmer`` keeps
        several searches in flight, and two of them may hold the same HMM object."""
        L = len(sequences[0]) if len(sequences) else self.L_HINT
        cfg = self._cfg()
        om = self._windowed_om(query, min(L, 100000), cfg)
        if self.bit_cutoffs is not None and not getattr(om.cutoffs, self.bit_cutoffs + "_available")():
            raise MissingCutoffs(om.name, self.bit_cutoffs)
        return om, cfg

    def search_hmm(self, query, sequences, devices: Optional[Sequence[int]] = None, _prepared=None) -> "TopHits":
        """``nhmmer`` with ``query`` against the long targets of ``sequences`` (reference ``plan7.pyx:7272-7418``).

        ``devices``: deal the (target, block, strand) units of the search -- the iterations of the reference's loop
        (``plan7.pyx:7582-7655``), which are independent of one another -- over these devices, consecutive units per
        device, one host thread each; the parts are finished together (E-values for all residues searched, duplicate
        removal across block boundaries, thresholds), so the result is the one-device result."""
        if isinstance(sequences, SequenceFile):
            if sequences.name is None:
                raise ValueError("can only use a `SequenceFile` backed by a file for reading targets")
            if not sequences.digital:
                raise ValueError("target sequences file is not in digital mode")
            sequences = sequences.read_block()
        if query.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, query.alphabet)
        if sequences.alphabet != self.alphabet:
            raise AlphabetMismatch(self.alphabet, sequences.alphabet)
        if isinstance(query, (Profile, OptimizedProfile)) and self.window_length is None and (getattr(query, "max_length", None) or -1) <= 0:
            raise TypeError("Cannot use `Profile` or `OptimizedProfile` query without `max_length` set")     # plan7.pyx:7354
        if _prepared is None:
            _prepared = self._prepare_query(query, sequences)
        om, cfg = _prepared
        if isinstance(sequences, DigitalSequenceBlock) and all(len(s) < 2 ** 31 for s in sequences):
            # the block's cached flat image (the same "255 x1..xL 255 ..." layout): a chromosome is not copied per query
            pk = sequences.packed()
            n = len(sequences)
            dsq, offsets, lengths = pk.dsq, pk.offsets, pk.lengths.astype(np.int64)
            # ... nor uploaded per query: the image is immutable (a mutated block gets a new one), its token lets the device
            # keep its copy between searches (cfg.lt_resident_key; one copy per device)
            with _RESIDENT_LOCK:                # one token and one finalizer per image, whoever asks first
                tok = getattr(pk, "_resident_token", None)
                if tok is None:
                    tok = next(_RESIDENT_TOKENS)
                    try:
                        pk._resident_token = tok
                        # the device copies go when the image does (p7x_longtargets_release_resident: every device, this key)
                        weakref.finalize(pk, _release_resident, tok)
                    except (AttributeError, TypeError):
                        tok = 0
            cfg.lt_resident_key = tok
            names = (C.c_char_p * max(n, 1))(*[s.name.encode() for s in sequences])
            accs = (C.c_char_p * max(n, 1))(*[(s.accession or "").encode() for s in sequences])
            descs = (C.c_char_p * max(n, 1))(*[(s.description or "").encode() for s in sequences])
        else:
            dsq, offsets, lengths, names, accs, descs = self._pack(sequences)

        def part(device: int, k: int, nparts: int) -> C.c_void_p:
            c = _lib.PipelineCfg.from_buffer_copy(cfg)
            c.lt_part, c.lt_nparts = k, nparts
            if nparts > 1:                              # the parts run side by side: each gets its share of the host workers
                usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                c.host_threads = max(1, (c.host_threads if c.host_threads > 0 else usable) // nparts)
            out = C.c_void_p()
            st = _lib.lib().p7x_search_longtargets(C.byref(c), om._handle, device, dsq.ctypes.data, offsets.ctypes.data,
                                                   lengths.ctypes.data, len(sequences), names, accs, descs, C.byref(out))
            if st != 0:
                raise status_to_exception(st, "p7x_search_longtargets", _lib.last_error())
            return out

        devices = list(devices) if devices else [self.device]
        if len(devices) == 1:
            out = part(devices[0], 0, 1)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(devices)) as ex:
                futs = [ex.submit(part, d, k, len(devices)) for k, d in enumerate(devices)]
                handles, err = [], None
                for f in futs:
                    try:
                        handles.append(f.result())
                    except BaseException as e:          # noqa: BLE001 - the other parts are still released below
                        err = err or e
            if err is not None:
                for h in handles:
                    _lib.lib().p7x_tophits_destroy(h)
                raise err
            arr = (C.c_void_p * len(handles))(*[h.value for h in handles])
            out = C.c_void_p()
            st = _lib.lib().p7x_tophits_merge_longtargets(arr, len(handles), C.byref(out))      # consumes the parts
            if st != 0:
                raise status_to_exception(st, "p7x_tophits_merge_longtargets", _lib.last_error())
        hits = TopHits(query, out)
        hits._om = om
        return hits


_RESIDENT_TOKENS = itertools.count(1)          # one per packed image whose device copy may be kept (cfg.lt_resident_key)
_RESIDENT_LOCK = threading.Lock()


def _release_resident(token: int) -> None:
    try:
        _lib.lib().p7x_longtargets_release_resident(-1, token)
    except Exception:                           # noqa: BLE001 - interpreter shutdown: the library may be gone
        pass


class SequenceDatabase:
    """A ``DigitalSequenceBlock`` packed once and resident in the HBM of one device (``p7x_seqdb``);
    any number of queries can then be searched against it without re-uploading the targets."""

    def __init__(self, block: DigitalSequenceBlock, device: int = 0):
        self.block = block
        self.device = device
        self.alphabet = block.alphabet
        limit = 100000                                            # plan7.pyx:5421, 6218-6219
        pk = block.packed()
        if pk.n and int(pk.lengths.max()) > limit:
            raise ValueError(f"sequence length over comparison pipeline limit ({limit})")
        self._handle = C.c_void_p()
        st = _lib.lib().p7x_seqdb_create(device, block.alphabet.type_code, pk.dsq.ctypes.data, pk.offsets.ctypes.data,
                                         pk.lengths.ctypes.data, pk.n, C.byref(self._handle))
        if st != 0:
            raise status_to_exception(st, "p7x_seqdb_create", _lib.last_error())
        n = pk.n
        lazy = getattr(block, "_list", 0) is None          # packed-only block from the native FASTA parser
        if lazy:
            # name / description tables: arrays of char* into the parser's NUL-terminated string table
            base = block._strtab.ctypes.data
            self._name_ptrs = (block._name_off + base).astype(np.uint64)
            self._desc_ptrs = (block._desc_off + base).astype(np.uint64)
            self._names = C.c_void_p(self._name_ptrs.ctypes.data)
            self._descs = C.c_void_p(self._desc_ptrs.ctypes.data)
            self._accs = None
        else:
            self._names = (C.c_char_p * max(n, 1))(*[s.name.encode() for s in block])
            self._accs = (C.c_char_p * max(n, 1))(*[(s.accession or "").encode() for s in block])
            self._descs = (C.c_char_p * max(n, 1))(*[(s.description or "").encode() for s in block])

    @classmethod
    def from_packed(cls, alphabet: Alphabet, dsq: np.ndarray, offsets: np.ndarray, lengths: np.ndarray,
                    device: int = 0, names=None) -> "SequenceDatabase":
        """Build directly from flat arrays (the C-ABI's own input format), skipping per-sequence Python objects.
        Hits then carry the target index (``Hit.seqidx``) and, if ``names`` is None, an empty name."""
        self = cls.__new__(cls)
        self.block, self.device = None, device
        self._n = int(lengths.shape[0])
        if self._n and int(lengths.max()) > 100000:
            raise ValueError("sequence length over comparison pipeline limit (100000)")
        self._keep = (np.ascontiguousarray(dsq, dtype=np.uint8), np.ascontiguousarray(offsets, dtype=np.int64),
                      np.ascontiguousarray(lengths, dtype=np.int32))
        self.alphabet = alphabet
        self._handle = C.c_void_p()
        st = _lib.lib().p7x_seqdb_create(device, alphabet.type_code, self._keep[0].ctypes.data, self._keep[1].ctypes.data,
                                         self._keep[2].ctypes.data, self._n, C.byref(self._handle))
        if st != 0:
            raise status_to_exception(st, "p7x_seqdb_create", _lib.last_error())
        if names is None:
            self._names = self._accs = self._descs = None
        else:
            self._names = (C.c_char_p * max(self._n, 1))(*[n.encode() for n in names])
            self._accs = self._descs = None
        return self

    def __len__(self) -> int:
        return len(self.block) if self.block is not None else self._n

    def __del__(self):
        if getattr(self, "_handle", None):
            try:
                _lib.lib().p7x_seqdb_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    def ensemble(self, om: OptimizedProfile, target: int, start: int, end: int, *, seed: int = 42, device: bool = True):
        """Test seam (``p7x_debug_ensemble``): the 200 stochastic tracebacks of region ``start..end`` (1-based) of one
        target, sampled by the device kernels or by the host twin.  Returns ``(status, domains[n, 5], null2_sums[Lr + 1])``:
        ``domains`` rows are (sample, first residue, last residue, first node, last node), residues counted inside the region."""
        Lr = end - start + 1
        cap = 200 * 16
        ndom, status = C.c_int32(0), C.c_int32(0)
        dom = np.zeros((cap, 5), dtype=np.int32)
        n2 = np.zeros(Lr + 1, dtype=np.float32)
        st = _lib.lib().p7x_debug_ensemble(om._handle, self._handle, int(target), int(start), int(end), int(seed), int(bool(device)),
                                           C.byref(ndom), dom.ctypes.data, cap, n2.ctypes.data, C.byref(status))
        if st != 0:
            raise status_to_exception(st, "p7x_debug_ensemble", _lib.last_error())
        return status.value, dom[:min(ndom.value, cap)].copy(), n2

    def filters(self, om: OptimizedProfile, msv=True, viterbi=False, forward=False, bias=False):
        """Raw per-target filter outputs (``p7x_filters_batch``): dict of numpy arrays in target order."""
        n = len(self)
        out = {}
        xj = np.zeros(n, dtype=np.int32) if msv else None
        xc = np.zeros(n, dtype=np.int32) if viterbi else None
        fw = np.zeros(n, dtype=np.float32) if forward else None
        bs = np.zeros(n, dtype=np.float32) if bias else None
        st = _lib.lib().p7x_filters_batch(om._handle, self._handle,
                                          None if xj is None else xj.ctypes.data, None if xc is None else xc.ctypes.data,
                                          None if fw is None else fw.ctypes.data, None if bs is None else bs.ctypes.data)
        if st != 0:
            raise status_to_exception(st, "p7x_filters_batch", _lib.last_error())
        if msv:
            out["xJ"] = xj
        if viterbi:
            out["xC"] = xc
        if forward:
            out["fwd"] = fw
        if bias:
            out["filtersc"] = bs
        return out
