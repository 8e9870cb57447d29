#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_config3.py -q -x 2>&1 | tail -2
