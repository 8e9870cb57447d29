"""Cycles per phase of the envelope kernel on the headline workload.  Needs a library whose p7x_envelope.hip was compiled
with -DP7X_ENV_PROFILE (scripts/env_variant.sh makes one and puts it in place of libp7x.so on the GPU box).
Usage: python scripts/env_phase_profile.py [bench.py flags]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pyhmmer_amd import _lib

sys.argv = ["bench.py", "--workload", "config1", "--steps", "2", "--warmup", "1"] + sys.argv[1:]
bench.main()
out = (C.c_ulonglong * 8)()
fn = _lib.lib().p7x_debug_env_profile
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(out) == 0
v = list(out)
rows, envs = max(1, v[4]), max(1, v[5])
tot = sum(v[:4])
print(f"envelopes {envs}, rows {rows} ({rows / envs:.1f} per envelope); wavefront cycles per envelope {tot / envs:.0f}")
for i, name in enumerate(("1 Forward", "2 Backward", "3 decoding + OA", "4 traceback")):
    print(f"  phase {name:18s} {v[i] / tot * 100:5.1f} %   {v[i] / rows:8.0f} cycles per row")
