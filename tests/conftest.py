import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def libp7x():
    """The product library.  Built by __graft_entry__.build(); tests never build it implicitly on the GPU box."""
    from pyhmmer_amd import _lib
    if not _lib.LIB_PATH.exists():
        _lib.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def proteome():
    from pyhmmer_amd import easel
    with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True,
                            alphabet=easel.Alphabet.amino()) as sf:
        return sf.read_block()


def load_hmms(name):
    from pyhmmer_amd import plan7
    with plan7.HMMFile(GOLDEN / "hmms" / f"{name}.hmm") as f:
        return list(f)


@pytest.fixture(scope="session")
def models(libp7x):
    out = {}
    for name in ("PF02826", "Thioesterase", "RREFam", "KR", "LuxC"):
        out[name] = load_hmms(name)
    return out


def golden_table(name, query=None, kind="tbl"):
    """Rows of a HMMER tabular output file (whitespace split; comment lines dropped)."""
    rows = []
    for line in open(GOLDEN / "tables" / name):
        if line.startswith("#") or not line.strip():
            continue
        f = line.split()
        qcol = 2 if kind == "tbl" else 3
        if query is None or f[qcol] == query:
            rows.append(f)
    return rows


def synthetic_block(n, L, seed, alphabet=None, lengths=None):
    """i.i.d. background sequences (SURVEY.md 8d config 2 recipe)."""
    import numpy as np
    from pyhmmer_amd import easel, plan7
    abc = alphabet or easel.Alphabet.amino()
    bg = plan7.Background(abc)
    rng = np.random.default_rng(seed)
    p = bg.residue_frequencies.astype(np.float64)
    p /= p.sum()
    seqs = []
    for t in range(n):
        Lt = L if lengths is None else int(lengths[t])
        seqs.append(easel.DigitalSequence(abc, name=f"syn{t}", sequence=rng.choice(abc.K, size=Lt, p=p).astype(np.uint8)))
    return easel.DigitalSequenceBlock(abc, seqs)
