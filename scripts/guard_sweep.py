"""Calibration of the two near-tie guards on the GPU box: the device path (hmmer.hmmsearch, defaults) against the oracle's
domain definition (upstream's summation order) for a sweep of cfg.oa_guard and cfg.ens_guard.  Per setting: envelopes /
ensemble targets compared, how many differ in any coordinate, how many units the guard sent back to the host stage."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import oracle_lib as oracle                          # noqa: E402
from conftest import load_hmms                       # noqa: E402
from test_oracle_domains import _homolog_block       # noqa: E402
from pyhmmer_amd import hmmer, plan7                 # noqa: E402


def main():
    loose = dict(E=1e9, domE=1e9, incE=1e9, incdomE=1e9)
    cases = []
    for model in ("PF02826", "KR", "LuxC", "Thioesterase"):
        hmm = load_hmms(model)[0]
        for seed in (21, 22, 23):
            block = _homolog_block(hmm, 20, 300, seed=seed)
            op = oracle.OracleProfile(hmm, plan7.Background(hmm.alphabet), 400)
            want = {}
            for s in block:
                if s.name.startswith("hom"):
                    envs, counts = oracle.domains(op, np.asarray(s.sequence, dtype=np.uint8))
                    want[s.name] = ([tuple(int(v) for v in e[:6]) for e in envs], counts)
            cases.append((model, seed, hmm, block, want))
    print("cases ready", len(cases), flush=True)

    def run(oa, ens):
        tot = dict(single=0, single_diff=0, ens=0, ens_diff=0, oa_redone=0, ens_device=0, ens_redone=0)
        for model, seed, hmm, block, want in cases:
            hits = next(iter(hmmer.hmmsearch(hmm, block, oa_guard=oa, ens_guard=ens, **loose)))
            g = hits.guard_counts
            tot["oa_redone"] += g["oa_redone"]; tot["ens_device"] += g["ens_device"]; tot["ens_redone"] += g["ens_redone"]
            for h in hits:
                if h.name not in want:
                    continue
                theirs, counts = want[h.name]
                ours = [(d.env_from, d.env_to, d.alignment.target_from, d.alignment.target_to, d.alignment.hmm_from, d.alignment.hmm_to) for d in h.domains]
                if counts[2] == 0:
                    tot["single"] += len(theirs)
                    tot["single_diff"] += sum(1 for o, t in zip(ours, theirs) if o != t) + abs(len(ours) - len(theirs))
                else:
                    tot["ens"] += 1
                    tot["ens_diff"] += ours != theirs
        return tot

    for oa in (0.0, 5e-7, 1e-6, 2e-6, 4e-6, 8e-6):
        print(f"oa_guard {oa:8.1e} ens_guard 2.5e-07:", run(oa, 2.5e-7), flush=True)
    for ens in (0.0, 3e-8, 6e-8, 1.2e-7, 2.5e-7, 5e-7, 1e-6):
        print(f"oa_guard 4.0e-06 ens_guard {ens:8.1e}:", run(4e-6, ens), flush=True)


if __name__ == "__main__":
    main()
