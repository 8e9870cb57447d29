#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in $1; do scripts/env_variant.sh $o -- python scripts/env_config1.py $o 2>&1 | tail -1; done
