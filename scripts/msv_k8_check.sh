#!/bin/bash
# the eight-lane MSV tiles (M 1022..2048): parity against the oracle, then the kernel's rate by model length
python -m pytest tests/test_gpu_filters.py -m gpu -x -q -k "msv" 2>&1 | tail -5 > gpurun_out/k8_tests.log
cat gpurun_out/k8_tests.log
python scripts/msv_by_length.py 262 1000 1021 1100 1500 2000 2048 2100 > gpurun_out/k8_rates.txt 2>&1
cat gpurun_out/k8_rates.txt
