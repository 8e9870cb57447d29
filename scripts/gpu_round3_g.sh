#!/bin/bash
# envelope kernel: rolled node loops (row state in scratch, more wavefronts) against unrolled + spilled, by model length
mkdir -p gpurun_out/r3g
echo "== unrolled up to 32 nodes per lane (default)" > gpurun_out/r3g/env_by_length.txt
timeout 600 python scripts/env_by_length.py 1000 1280 1536 2000 3000 5000 >> gpurun_out/r3g/env_by_length.txt 2>&1
for mx in 16 12; do
  touch pyhmmer_amd/csrc/p7x_envelope.hip
  P7X_CXXFLAGS="-DP7X_ENV_UNROLL_MAX=$mx" python -c "from pyhmmer_amd import _lib; _lib.build()" >> gpurun_out/r3g/build.log 2>&1
  echo "== unrolled up to $mx nodes per lane" >> gpurun_out/r3g/env_by_length.txt
  timeout 600 python scripts/env_by_length.py 768 1000 1280 1536 2000 3000 >> gpurun_out/r3g/env_by_length.txt 2>&1
done
cat gpurun_out/r3g/env_by_length.txt
