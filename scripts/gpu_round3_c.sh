#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_envelopes.py -x -q 2>&1 | tail -3
cp pyhmmer_amd/libp7x.so /tmp/keep.so
scripts/gpu_ab_env3.sh "$1"
cp /tmp/keep.so pyhmmer_amd/libp7x.so
