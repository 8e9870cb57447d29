"""Run the fast and the exact MSV kernels twice each on the bench workload and compare every xJ with the oracle."""
import os, subprocess, sys
import numpy as np
CODE = r'''
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bench
from conftest import load_hmms
from pyhmmer_amd import plan7
hmm = load_hmms("KR")[0]
flat, off, ln, planted = bench.make_workload(hmm, 1000000, 300, 42)
db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, off, ln)
om = plan7.OptimizedProfile(hmm, plan7.Background(hmm.alphabet), 300)
for rep in range(3):
    np.save(sys.argv[1] + f"_{rep}.npy", db.filters(om, msv=True)["xJ"])
if sys.argv[2] == "oracle":
    import oracle_lib
    from types import SimpleNamespace
    pk = SimpleNamespace(dsq=flat, offsets=off, lengths=ln, n=len(ln))
    np.save("/tmp/xj_oracle.npy", oracle_lib.OracleProfile(hmm, plan7.Background(hmm.alphabet), 300).msv_block(pk))
'''
for tag, env in (("fast", {}), ("exact", {"P7X_MSV_EXACT": "1"})):
    subprocess.run([sys.executable, "-c", CODE, f"/tmp/xj_{tag}", "oracle" if tag == "fast" else "-"], check=True,
                   env={**os.environ, **env})
want = np.load("/tmp/xj_oracle.npy")
for tag in ("fast", "exact"):
    for rep in range(3):
        got = np.load(f"/tmp/xj_{tag}_{rep}.npy")
        bad = np.nonzero(got != want)[0]
        print(tag, rep, "mismatches vs oracle:", bad.size, bad[:8], got[bad[:8]], want[bad[:8]])
