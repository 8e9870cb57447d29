#!/bin/bash
# gpu_knobs.sh under an 8-CPU affinity mask (what a rank of an 8-GPU node with 64 cores would have)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
exec taskset -c 0-7 bash scripts/gpu_knobs.sh
