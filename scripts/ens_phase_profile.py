"""Cycles per phase of the ensemble walk kernel on the headline workload.  Needs a library whose p7x_ensemble.hip was compiled
with -DP7X_ENS_PROFILE (scripts/ens_variant.sh puts one in place of libp7x.so on the GPU box)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pyhmmer_amd import _lib

sys.argv = ["bench.py", "--workload", "config1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--spinup-max", "1"] + sys.argv[1:]
bench.main()
out = (C.c_ulonglong * 8)()
fn = _lib.lib().p7x_debug_ens_profile
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(out) == 0
v = list(out)
regions, samples, steps = max(1, v[7]), max(1, v[5]), max(1, v[4])
tot = sum(v[:4])
print(f"regions {regions}, samples {samples}, emitting core steps {steps} ({steps / samples:.1f} per sample), cache misses {v[6]} ({v[6] / steps * 100:.1f} % of steps)")
print(f"wavefront cycles per region {tot / regions:.0f}, per sample {tot / samples:.0f}")
for i, name in enumerate(("C/J runs", "select_e", "core walk", "domain finish")):
    print(f"  {name:14s} {v[i] / tot * 100:5.1f} %   {v[i] / samples:9.0f} cycles per sample")
print(f"  core walk: {v[2] / steps:.0f} cycles per emitting step")
