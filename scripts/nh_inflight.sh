#!/bin/bash
out=gpurun_out/nh_inflight.txt; : > $out
for k in 4 6 8; do
python - <<PY >> $out 2>/dev/null
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench, bench_workloads as bw
from pyhmmer_amd import plan7, hmmer, easel
sys.path.insert(0, "tests")
from conftest import load_hmms
hmm = load_hmms("bmyD")[0]
seq, planted = bw.make_chromosome(hmm, int(250e6), 50) if hasattr(bw, "make_chromosome") else (None, None)
PY
done
cat $out
