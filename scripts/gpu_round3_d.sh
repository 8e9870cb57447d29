#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in $1; do
  echo "== $o"
  scripts/env_variant.sh $o -- python -m pytest "tests/test_gpu_envelopes.py::test_device_envelopes_for_every_kernel_instantiation" -q 2>&1 | tail -4
done
