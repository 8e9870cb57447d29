import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "oracle"))          # oracle_lib: the checker's ctypes binding lives with the checker
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


def random_hmm(M, seed, alphabet=None, conserved=0.6):
    """A synthetic core model of any length (test input only): Dirichlet emissions sharpened towards one residue per
    node, HMMER's boundary conventions for nodes 0 and M (p7_hmm.c: t[0] has no D state, node M has no MD/DD)."""
    import numpy as np
    from pyhmmer_amd import easel, plan7
    abc = alphabet or easel.Alphabet.amino()
    K = abc.K
    rng = np.random.default_rng(seed)
    hmm = plan7.HMM(abc, M, f"rnd{M}_{seed}")
    bgf = plan7.Background(abc).residue_frequencies.astype(np.float64)
    mat = rng.dirichlet(np.full(K, 0.4), size=M + 1)
    peak = rng.integers(0, K, size=M + 1)
    mat = (1.0 - conserved) * mat
    mat[np.arange(M + 1), peak] += conserved
    mat[0] = 0.0
    mat[0, 0] = 1.0
    ins = np.tile(bgf / bgf.sum(), (M + 1, 1))
    t = np.zeros((M + 1, 7))
    mm = rng.uniform(0.90, 0.98, size=M + 1)
    mi = (1.0 - mm) * rng.uniform(0.3, 0.7, size=M + 1)
    t[:, 0], t[:, 1], t[:, 2] = mm, mi, 1.0 - mm - mi
    t[:, 3] = rng.uniform(0.4, 0.8, size=M + 1)
    t[:, 4] = 1.0 - t[:, 3]
    t[:, 5] = rng.uniform(0.5, 0.9, size=M + 1)
    t[:, 6] = 1.0 - t[:, 5]
    t[0, 5], t[0, 6] = 1.0, 0.0
    t[M, 0], t[M, 2] = t[M, 0] + t[M, 2], 0.0
    t[M, 5], t[M, 6] = 1.0, 0.0
    hmm.transition_probabilities[:] = t
    hmm.match_emissions[:] = mat
    hmm.insert_emissions[:] = ins
    hmm.composition = (mat[1:].mean(axis=0)).astype(np.float32)
    hmm.consensus = "".join(abc.symbols[int(p)].lower() for p in peak[1:])
    hmm._evparam[:] = [-9.0, 0.69, -10.0, 0.69, -4.0, 0.69]
    hmm.max_length = 4 * M
    return hmm
