#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py tests/test_gpu_connected_step.py tests/test_gpu_reference_callers.py -q -p no:cacheprovider -x 2>&1 | tail -2
D2AMD_POOL_STEAL=2 timeout 600 python -m pytest tests/test_gpu_pooler.py -q -p no:cacheprovider -x 2>&1 | tail -2
for rep in 1 2; do timeout 120 python scripts/pool_bwd_ab.py head 2>&1 | tail -1; done
