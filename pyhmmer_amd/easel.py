"""Minimal host-side mirror of the parts of ``pyhmmer.easel`` that sit on the
``p7_Pipeline`` hot path: the digital alphabets, digital sequences, the
``DigitalSequenceBlock`` input container and a FASTA ``SequenceFile`` reader.

Reference: ``src/pyhmmer/easel.pyx`` -- ``Alphabet`` (:183-528), ``TextSequence``
(:7483), ``DigitalSequence`` (:7741), ``DigitalSequenceBlock`` (:8629-8665),
``SequenceFile`` (read / read_block).  Everything else in ``pyhmmer.easel``
(MSA, SSI, matrices, genetic codes, ...) is out of scope.

The one deliberate difference from the reference: a ``DigitalSequenceBlock`` here can
be *packed* once (``block.packed()``) into the flat ``dsq_concat + offsets + lengths``
arrays the C-ABI consumes (``include/p7x.h``), instead of an array of ``ESL_SQ*``.
"""
from __future__ import annotations

import io
import os
from typing import Iterable, Iterator, List, Optional, Sequence as _Seq

import numpy as np

__all__ = [
    "Alphabet", "Sequence", "TextSequence", "DigitalSequence",
    "TextSequenceBlock", "DigitalSequenceBlock", "SequenceFile",
]

_AMINO = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"
_DNA = "ACGT-RYMKSWHBVDN*~"
_RNA = "ACGU-RYMKSWHBVDN*~"

eslRNA, eslDNA, eslAMINO = 1, 2, 3   # libeasel/alphabet.pxd type codes
DSQ_SENTINEL = 255                   # libeasel/__init__.pxd:21 (eslDSQ_SENTINEL)


class Alphabet:
    """A biological alphabet (reference ``easel.pyx:183-528``)."""

    __slots__ = ("type_code", "symbols", "K", "Kp", "_lut")

    def __init__(self, type_code: int):
        if type_code == eslAMINO:
            self.symbols, self.K = _AMINO, 20
        elif type_code == eslDNA:
            self.symbols, self.K = _DNA, 4
        elif type_code == eslRNA:
            self.symbols, self.K = _RNA, 4
        else:
            raise ValueError(f"unsupported alphabet type {type_code}")
        self.type_code = type_code
        self.Kp = len(self.symbols)
        lut = np.full(256, 255, dtype=np.uint8)
        for i, c in enumerate(self.symbols):
            lut[ord(c)] = i
            lut[ord(c.lower())] = i
        if type_code != eslAMINO:          # Easel synonyms: U<->T, X -> N, I->A? (only the common ones)
            lut[ord("U")] = lut[ord("u")] = 3
            lut[ord("T")] = lut[ord("t")] = 3
            lut[ord("X")] = lut[ord("x")] = self.symbols.index("N")
        lut[ord("_")] = lut[ord(".")] = self.K   # gap synonyms
        self._lut = lut

    # -- constructors mirroring the reference class methods
    @classmethod
    def amino(cls) -> "Alphabet":
        return cls(eslAMINO)

    @classmethod
    def dna(cls) -> "Alphabet":
        return cls(eslDNA)

    @classmethod
    def rna(cls) -> "Alphabet":
        return cls(eslRNA)

    @property
    def type(self) -> str:
        return {eslAMINO: "amino", eslDNA: "DNA", eslRNA: "RNA"}[self.type_code]

    @property
    def gap_symbol(self) -> str:
        return self.symbols[self.K]

    @property
    def gap_index(self) -> int:
        return self.K

    def is_amino(self) -> bool:
        return self.type_code == eslAMINO

    def is_dna(self) -> bool:
        return self.type_code == eslDNA

    def is_rna(self) -> bool:
        return self.type_code == eslRNA

    def is_nucleotide(self) -> bool:
        return self.type_code in (eslDNA, eslRNA)

    def __eq__(self, other) -> bool:
        return isinstance(other, Alphabet) and other.type_code == self.type_code

    def __hash__(self) -> int:
        return hash(self.type_code)

    def __repr__(self) -> str:
        return f"Alphabet.{ {eslAMINO: 'amino', eslDNA: 'dna', eslRNA: 'rna'}[self.type_code] }()"

    def encode(self, sequence: str) -> np.ndarray:
        raw = np.frombuffer(sequence.encode("ascii"), dtype=np.uint8)
        enc = self._lut[raw]
        if (enc == 255).any():
            bad = sequence[int(np.argmax(enc == 255))]
            raise ValueError(f"Invalid symbol {bad!r} for alphabet {self!r}")
        return enc

    def decode(self, sequence) -> str:
        arr = np.asarray(sequence, dtype=np.uint8)
        if arr.size and int(arr.max()) >= self.Kp:
            raise ValueError("invalid digital code in sequence")
        return "".join(self.symbols[i] for i in arr.tolist())


class Sequence:
    """Abstract biological sequence with metadata (reference ``easel.pyx`` ``Sequence``)."""

    __slots__ = ("name", "description", "accession", "source")

    def __init__(self, name: str = "", description: str = "", accession: str = "", source: str = ""):
        self.name = name
        self.description = description
        self.accession = accession
        self.source = source


class TextSequence(Sequence):
    """A sequence stored as text (reference ``easel.pyx:7483``)."""

    __slots__ = ("sequence",)

    def __init__(self, name: str = "", description: str = "", accession: str = "",
                 sequence: str = "", source: str = ""):
        super().__init__(name, description, accession, source)
        self.sequence = sequence

    def __len__(self) -> int:
        return len(self.sequence)

    def digitize(self, alphabet: Alphabet) -> "DigitalSequence":
        return DigitalSequence(alphabet, name=self.name, description=self.description,
                               accession=self.accession, sequence=alphabet.encode(self.sequence),
                               source=self.source)

    def copy(self) -> "TextSequence":
        return TextSequence(self.name, self.description, self.accession, self.sequence, self.source)


class DigitalSequence(Sequence):
    """A sequence stored as digital residue codes 0..Kp-1 (reference ``easel.pyx:7741``)."""

    __slots__ = ("alphabet", "sequence")

    def __init__(self, alphabet: Alphabet, name: str = "", description: str = "", accession: str = "",
                 sequence=None, source: str = ""):
        super().__init__(name, description, accession, source)
        self.alphabet = alphabet
        seq = np.zeros(0, dtype=np.uint8) if sequence is None else np.ascontiguousarray(sequence, dtype=np.uint8)
        if seq.size and int(seq.max()) >= alphabet.Kp:
            raise ValueError("invalid digital code in sequence")
        self.sequence = seq

    def __len__(self) -> int:
        return int(self.sequence.shape[0])

    def textize(self) -> TextSequence:
        return TextSequence(self.name, self.description, self.accession,
                            self.alphabet.decode(self.sequence), self.source)

    def copy(self) -> "DigitalSequence":
        return DigitalSequence(self.alphabet, self.name, self.description, self.accession,
                               self.sequence.copy(), self.source)


class PackedBlock:
    """Flat, device-friendly image of a ``DigitalSequenceBlock``.

    ``dsq`` holds ``255 x1..xL 255 x1..xL 255 ...`` (Easel sentinels, ``plan7.pyx:7615-7616``);
    ``offsets[t]`` indexes ``x1`` of target ``t``; ``lengths[t]`` is ``L_t``.
    """

    __slots__ = ("dsq", "offsets", "lengths", "n", "total_residues", "_resident_token", "__weakref__")

    def __init__(self, seqs: _Seq[DigitalSequence]):
        n = len(seqs)
        lengths = np.fromiter((len(s) for s in seqs), dtype=np.int32, count=n)
        offsets = np.empty(n, dtype=np.int64)
        total = int(lengths.sum(dtype=np.int64))
        dsq = np.full(total + n + 1, DSQ_SENTINEL, dtype=np.uint8)
        pos = 1
        for t, s in enumerate(seqs):
            L = int(lengths[t])
            offsets[t] = pos
            dsq[pos:pos + L] = s.sequence
            pos += L + 1
        self.dsq, self.offsets, self.lengths = dsq, offsets, lengths
        self.n, self.total_residues = n, total

    @classmethod
    def from_arrays(cls, dsq: np.ndarray, offsets: np.ndarray, lengths: np.ndarray) -> "PackedBlock":
        self = cls.__new__(cls)
        self.dsq, self.offsets, self.lengths = dsq, offsets, lengths
        self.n, self.total_residues = int(lengths.shape[0]), int(lengths.sum(dtype=np.int64))
        return self


class _SequenceBlock:
    __slots__ = ("_seqs", "_packed", "_version")

    def __init__(self, iterable: Iterable = ()):
        self._seqs: List = list(iterable)
        self._packed = None
        self._version = 0           # bumped by every mutation: caches of the packed / resident copy key on it

    def _touch(self) -> None:
        self._packed = None
        self._version += 1

    def __len__(self) -> int:
        return len(self._seqs)

    def __iter__(self) -> Iterator:
        return iter(self._seqs)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return type(self)._from_list(self, self._seqs[index])
        return self._seqs[index]

    def __setitem__(self, index, value) -> None:
        self._seqs[index] = value
        self._touch()

    def __delitem__(self, index) -> None:
        del self._seqs[index]
        self._touch()

    def __contains__(self, item) -> bool:
        return item in self._seqs

    def append(self, seq) -> None:
        self._seqs.append(seq)
        self._touch()

    def extend(self, iterable) -> None:
        self._seqs.extend(iterable)
        self._touch()

    def insert(self, index: int, seq) -> None:
        self._seqs.insert(index, seq)
        self._touch()

    def pop(self, index: int = -1):
        seq = self._seqs.pop(index)
        self._touch()
        return seq

    def remove(self, seq) -> None:
        self._seqs.remove(seq)
        self._touch()

    def index(self, seq, start: int = 0, stop: Optional[int] = None) -> int:
        return self._seqs.index(seq, start, len(self._seqs) if stop is None else stop)

    def clear(self) -> None:
        self._seqs.clear()
        self._touch()

    def largest(self):
        if not self._seqs:
            raise ValueError("block is empty")
        return max(self._seqs, key=len)

    def total_length(self) -> int:
        return sum(len(s) for s in self._seqs)


class TextSequenceBlock(_SequenceBlock):
    """Reference ``easel.pyx:8501``."""

    @staticmethod
    def _from_list(parent, lst):
        return TextSequenceBlock(lst)

    def digitize(self, alphabet: Alphabet) -> "DigitalSequenceBlock":
        return DigitalSequenceBlock(alphabet, (s.digitize(alphabet) for s in self._seqs))


class DigitalSequenceBlock(_SequenceBlock):
    """A list of digital sequences searched as one batch (reference ``easel.pyx:8629-8665``)."""

    __slots__ = ("alphabet",)

    def __init__(self, alphabet: Alphabet, iterable: Iterable = ()):
        super().__init__(iterable)
        self.alphabet = alphabet
        for s in self._seqs:
            if not isinstance(s, DigitalSequence):
                raise TypeError(f"expected DigitalSequence, found {type(s).__name__}")
            if s.alphabet != alphabet:
                raise ValueError("alphabet mismatch in DigitalSequenceBlock")

    @staticmethod
    def _from_list(parent, lst):
        return DigitalSequenceBlock(parent.alphabet, lst)

    def copy(self) -> "DigitalSequenceBlock":
        return DigitalSequenceBlock(self.alphabet, (s.copy() for s in self._seqs))

    def packed(self) -> PackedBlock:
        """Pack once into the flat arrays of the C-ABI (cached until the block is mutated)."""
        if self._packed is None or self._packed.n != len(self._seqs):
            self._packed = PackedBlock(self._seqs)
        return self._packed


class _LazyDigitalSequenceBlock(DigitalSequenceBlock):
    """A block that exists only as the packed arrays the native FASTA parser produced (``p7x_fasta_parse``): a
    million-record file becomes a searchable block without a million Python objects.  ``DigitalSequence`` objects are
    built on demand (indexing) or all at once when the block is iterated or mutated."""

    __slots__ = ("_list", "_pk", "_strtab", "_name_off", "_desc_off")

    def __init__(self, alphabet: Alphabet, pk: PackedBlock, strtab: np.ndarray, name_off: np.ndarray, desc_off: np.ndarray):
        self._list = None
        self._pk, self._strtab, self._name_off, self._desc_off = pk, strtab, name_off, desc_off
        self.alphabet = alphabet
        self._packed = pk
        self._version = 0

    def _cstr(self, off: int) -> str:
        buf = self._strtab
        end = off
        n = buf.shape[0]
        while end < n and buf[end] != 0:
            end += 1
        return bytes(buf[off:end]).decode("utf-8", "replace")

    def _make(self, t: int) -> DigitalSequence:
        pk = self._pk
        o, L = int(pk.offsets[t]), int(pk.lengths[t])
        return DigitalSequence(self.alphabet, name=self._cstr(int(self._name_off[t])),
                               description=self._cstr(int(self._desc_off[t])), sequence=pk.dsq[o:o + L])

    @property
    def _seqs(self):
        if self._list is None:
            self._list = [self._make(t) for t in range(self._pk.n)]
        return self._list

    @_seqs.setter
    def _seqs(self, value):
        self._list = value

    def __len__(self) -> int:
        return self._pk.n if self._list is None else len(self._list)

    def __getitem__(self, index):
        if self._list is None and not isinstance(index, slice):
            n = self._pk.n
            i = index + n if index < 0 else index
            if not 0 <= i < n:
                raise IndexError("block index out of range")
            return self._make(i)
        return super().__getitem__(index)

    def total_length(self) -> int:
        return self._pk.total_residues if self._list is None else super().total_length()

    def packed(self) -> PackedBlock:
        return self._pk if self._list is None else super().packed()


class SequenceFile:
    """Sequence reader with the reference's ``SequenceFile`` surface (``read``, ``read_block``, iteration, context
    manager).  FASTA, and the sequence records of GenBank flat files (LOCUS / DEFINITION / VERSION / ORIGIN: what the
    reference's nhmmer fixtures need); the format is recognised from the first line when it is not given."""

    def __init__(self, file, format: Optional[str] = None, *, digital: bool = False,
                 alphabet: Optional[Alphabet] = None):
        if format not in (None, "fasta", "afa", "genbank"):
            raise ValueError(f"unsupported sequence format: {format!r}")
        if isinstance(file, (str, bytes, os.PathLike)):
            self._fh = open(file, "r")
            self._own = True
            self.name = os.fspath(file)
        else:
            self._fh = file
            self._own = False
            self.name = getattr(file, "name", None)
        self.digital = digital
        self.alphabet = alphabet
        self._pending: Optional[str] = None
        self._touched = False
        self._chunk_pos = 0
        self._chunking = False                                # read_chunk() is walking the file
        if format is None:
            try:
                pos = self._fh.tell()
                first = self._fh.readline()
                self._fh.seek(pos)
                format = "genbank" if first.startswith("LOCUS") else "fasta"
            except (OSError, ValueError):
                format = "fasta"
        self.format = format
        if digital and alphabet is None:
            self.alphabet = self.guess_alphabet()
            if self.alphabet is None:
                raise ValueError("Could not determine alphabet of file")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self) -> None:
        if self._own:
            self._fh.close()

    def rewind(self) -> None:
        self._fh.seek(0)
        self._pending = None
        self._touched = False
        self._chunk_pos = 0
        self._chunking = False

    def guess_alphabet(self) -> Optional[Alphabet]:
        pos = self._fh.tell()
        counts = np.zeros(256, dtype=np.int64)
        n = 0
        in_origin = self.format != "genbank"
        for line in self._fh:
            if line.startswith(">"):
                continue
            if self.format == "genbank":
                if line.startswith("ORIGIN"):
                    in_origin = True
                    continue
                if line.startswith("//"):
                    in_origin = False
                if not in_origin:
                    continue
                line = "".join(c for c in line if c.isalpha())
            raw = np.frombuffer(line.strip().upper().encode("ascii", "ignore"), dtype=np.uint8)
            counts += np.bincount(raw, minlength=256)
            n += raw.size
            if n > 4000:
                break
        self._fh.seek(pos)
        if n == 0:
            return None
        nuc = sum(int(counts[ord(c)]) for c in "ACGTUN")
        return Alphabet.dna() if nuc >= 0.9 * n else Alphabet.amino()

    def _read_genbank(self) -> Optional[TextSequence]:
        """One record of a GenBank flat file: name from LOCUS, accession from VERSION (else ACCESSION), description from
        DEFINITION (continuation lines joined), residues from ORIGIN .. //."""
        name = acc = ""
        desc: list = []
        chunks: list = []
        state = None
        seen = False
        for line in self._fh:
            if line.startswith("//"):
                if seen:
                    break
                continue
            key = line[:12].strip()
            if key == "LOCUS":
                name = line.split()[1]
                seen = True
                state = None
            elif key == "DEFINITION":
                desc = [line[12:].strip()]
                state = "def"
            elif key == "VERSION":
                parts = line.split()
                acc = parts[1] if len(parts) > 1 else acc
                state = None
            elif key == "ACCESSION":
                parts = line.split()
                acc = acc or (parts[1] if len(parts) > 1 else "")
                state = None
            elif key == "ORIGIN":
                state = "seq"
            elif state == "seq":
                chunks.append("".join(c for c in line if c.isalpha()))
            elif state == "def" and line.startswith(" "):
                desc.append(line.strip())
            elif key:
                state = None
        if not seen:
            return None
        return TextSequence(name=name, description=" ".join(desc), accession=acc, sequence="".join(chunks))

    def _read_text(self) -> Optional[TextSequence]:
        self._touched = True
        if self.format == "genbank":
            return self._read_genbank()
        header = self._pending
        self._pending = None
        if header is None:
            for line in self._fh:
                if line.startswith(">"):
                    header = line
                    break
            if header is None:
                return None
        chunks = []
        for line in self._fh:
            if line.startswith(">"):
                self._pending = line
                break
            chunks.append(line.strip())
        head = header[1:].rstrip("\r\n")
        parts = head.split(None, 1)
        name = parts[0] if parts else ""
        desc = parts[1].strip() if len(parts) > 1 else ""
        seq = "".join(chunks).replace(" ", "")
        return TextSequence(name=name, description=desc, sequence=seq)

    def read(self):
        s = self._read_text()
        if s is None:
            return None
        return s.digitize(self.alphabet) if self.digital else s

    def __iter__(self):
        return self

    def __next__(self):
        s = self.read()
        if s is None:
            raise StopIteration
        return s

    def _read_block_native(self) -> "DigitalSequenceBlock":
        """Whole file -> packed block through the C parser (no per-record Python work)."""
        block = self._parse_native(np.fromfile(self.name, dtype=np.uint8))
        self._touched = True
        self._fh.seek(0, 2)                                   # the file is consumed
        return block

    def read_chunk(self, max_bytes: int = 1 << 28) -> "DigitalSequenceBlock":
        """The next whole records of a digital FASTA file, about ``max_bytes`` of text at a time, through the C parser:
        how ``hmmsearch`` walks a target database that is larger than memory (the reference iterates the file, one
        sequence at a time: ``plan7.pyx:6244-6252``, ``_search_loop_file`` ``:6456``).  An empty block at the end of the
        file; ``rewind()`` starts over.  Other formats and text mode fall back to ``read_block(residues=max_bytes)``."""
        if not (self.digital and self._own and self.format != "genbank"):
            return self.read_block(residues=max_bytes)
        if self._touched and not self._chunking:
            raise ValueError("read_chunk() cannot continue a file that was partly read record by record; rewind() first")
        with open(self.name, "rb") as f:
            f.seek(self._chunk_pos)
            data = f.read(max_bytes)
            if len(data) == max_bytes:                        # most likely inside a record: take the rest of it
                tail = data[-1:]
                while True:
                    more = f.read(1 << 20)
                    if not more:
                        break
                    k = (tail + more).find(b"\n>")
                    if k >= 0:
                        data += more[:k]                      # up to and including the newline before the next header
                        break
                    data += more
                    tail = more[-1:]
        self._chunk_pos += len(data)
        self._touched = True
        self._chunking = True
        self._fh.seek(self._chunk_pos)                        # record-by-record reads continue where the chunks ended
        self._pending = None
        if not data.strip():
            return DigitalSequenceBlock(self.alphabet, [])
        return self._parse_native(np.frombuffer(data, dtype=np.uint8))

    def _parse_native(self, data: np.ndarray) -> "DigitalSequenceBlock":
        import ctypes as C
        from . import _lib
        lut = self.alphabet._lut.copy()
        for ch in b" \t\r\n0123456789":
            lut[ch] = 254                                     # ignored inside sequence data, as Easel's sqio does
        ns, nr, sb, bad = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        fn = _lib.lib().p7x_fasta_parse
        st = fn(data.ctypes.data, data.shape[0], lut.ctypes.data, C.byref(ns), C.byref(nr), C.byref(sb), None, None, None,
                None, None, None, C.byref(bad))
        if st == 0:
            dsq = np.empty(nr.value + ns.value + 1, dtype=np.uint8)
            offsets, lengths = np.empty(ns.value, dtype=np.int64), np.empty(ns.value, dtype=np.int32)
            strtab = np.empty(max(sb.value, 1), dtype=np.uint8)
            name_off, desc_off = np.empty(ns.value, dtype=np.int64), np.empty(ns.value, dtype=np.int64)
            st = fn(data.ctypes.data, data.shape[0], lut.ctypes.data, C.byref(ns), C.byref(nr), C.byref(sb), dsq.ctypes.data,
                    offsets.ctypes.data, lengths.ctypes.data, strtab.ctypes.data, name_off.ctypes.data, desc_off.ctypes.data,
                    C.byref(bad))
        if st != 0:
            raise ValueError(f"{self.name}: {_lib.last_error()} (byte {bad.value})")
        return _LazyDigitalSequenceBlock(self.alphabet, PackedBlock.from_arrays(dsq, offsets, lengths), strtab, name_off, desc_off)

    def read_block(self, sequences: Optional[int] = None, residues: Optional[int] = None):
        if self.digital and self._own and not self._touched and sequences is None and residues is None and self.format != "genbank":
            return self._read_block_native()
        out = []
        nres = 0
        while True:
            if sequences is not None and len(out) >= sequences:
                break
            if residues is not None and nres >= residues:
                break
            s = self.read()
            if s is None:
                break
            out.append(s)
            nres += len(s)
        if self.digital:
            return DigitalSequenceBlock(self.alphabet, out)
        return TextSequenceBlock(out)
