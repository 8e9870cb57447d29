"""Assertions against HMMER's tabular outputs (tests/golden/tables), shared by the CPU host tests and the GPU
end-to-end tests.  Tolerances are the reference's own (tests/test_hmmer.py:109-198: score/bias to 0.1 bit,
E-values equal at %9.2g) for every hit, including those whose null2 correction comes from the stochastic
traceback ensemble (nclustered > 0): the Easel LCG seeded with 42 reproduces upstream's draws."""
import itertools

import pytest

TOL_BITS = 0.1


def _same_evalue(got, want):
    if f"{got:9.2g}" == f"{want:9.2g}":
        return True
    return got == pytest.approx(want, rel=0.06)      # the table holds two significant digits


def check_tbl(hits, rows):
    reported = [h for h in hits if h.reported]
    assert len(reported) == len(rows)
    for row, hit in itertools.zip_longest(rows, reported):
        tol = TOL_BITS
        assert hit.name == row[0]
        assert hit.accession is None if row[1] == "-" else hit.accession == row[1]
        assert hit.score == pytest.approx(float(row[5]), abs=tol), hit.name
        assert hit.bias == pytest.approx(float(row[6]), abs=tol), hit.name
        assert _same_evalue(hit.evalue, float(row[4])), (hit.name, hit.evalue, row[4])
        assert hit.best_domain.score == pytest.approx(float(row[8]), abs=tol)
        assert hit.nexpected == pytest.approx(float(row[10]), abs=0.1)
        assert (hit.nregions, hit.nclustered, hit.noverlaps, hit.nenvelopes) == tuple(int(v) for v in row[11:15]), hit.name
        assert len(hit.domains) == int(row[15])
        assert len(hit.domains.reported) == int(row[16]) and len(hit.domains.included) == int(row[17]), hit.name


def check_domtbl(hits, rows):
    doms = [d for h in hits if h.reported for d in h.domains if d.reported]
    assert len(doms) == len(rows)
    for row, d in itertools.zip_longest(rows, doms):
        tol = TOL_BITS
        assert d.hit.name == row[0]
        assert d.score == pytest.approx(float(row[13]), abs=tol)
        assert d.bias == pytest.approx(float(row[14]), abs=tol)
        assert _same_evalue(d.c_evalue, float(row[11])), (d.hit.name, d.c_evalue, row[11])
        assert _same_evalue(d.i_evalue, float(row[12]))
        assert (d.alignment.hmm_from, d.alignment.hmm_to) == (int(row[15]), int(row[16]))
        assert (d.alignment.target_from, d.alignment.target_to) == (int(row[17]), int(row[18]))
        assert (d.env_from, d.env_to) == (int(row[19]), int(row[20]))
        assert d.accuracy == pytest.approx(float(row[21]), abs=0.01)
