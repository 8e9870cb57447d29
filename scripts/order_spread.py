"""Calibration of the ensemble walk's near-threshold guard (p7x_pipeline_cfg.ens_guard): how far apart the integer thresholds
of a stochastic traceback's choice points lie when the region's Forward matrix is summed in the device's lane-chunk order
and in upstream's striped order (p7x_debug_order_spread; host code, no GPU).  Regions: whole homolog targets, multihit."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_hmms, random_hmm          # noqa: E402
from test_oracle_domains import _homolog_block      # noqa: E402
from pyhmmer_amd import _lib, plan7                 # noqa: E402


def main():
    lib = _lib.lib()
    tot = np.zeros(12)
    for model in ("PF02826", "KR", "LuxC", "Thioesterase", 60, 150, 400, 900, 1500):
        hmm = load_hmms(model)[0] if isinstance(model, str) else random_hmm(model, seed=700 + model)
        om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 400)
        block = _homolog_block(hmm, 0, 40 if hmm.M < 800 else 10, seed=5)
        acc = np.zeros(12)
        for s in block:
            seq = np.concatenate([[255], np.asarray(s.sequence, dtype=np.uint8), [255]]).astype(np.uint8)
            out = (C.c_double * 12)()
            st = lib.p7x_debug_order_spread(om._handle, seq.ctypes.data, len(s), 1, len(s), 1, out)
            assert st == 0
            o = np.array(out[:])
            acc[0] += o[0]; acc[1] = max(acc[1], o[1]); acc[2:] += o[2:]
        print(f"{str(model):>14} M={hmm.M:5d} thresholds {acc[0]:.3g} max {acc[1]:.3g}  frac > 2^-k, k=24..15: " +
              " ".join(f"{v / acc[0]:.2e}" for v in acc[2:]))
        tot[0] += acc[0]; tot[1] = max(tot[1], acc[1]); tot[2:] += acc[2:]
    print(f"{'ALL':>14}         thresholds {tot[0]:.3g} max {tot[1]:.3g}  frac > 2^-k, k=24..15: " + " ".join(f"{v / tot[0]:.2e}" for v in tot[2:]))


if __name__ == "__main__":
    main()
