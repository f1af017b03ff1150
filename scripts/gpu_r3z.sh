#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
for seed in 1 2 3 4 5 6; do PYTHONHASHSEED=$seed timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -1; done
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_connected_step.py -q -p no:cacheprovider 2>&1 | tail -1; done
