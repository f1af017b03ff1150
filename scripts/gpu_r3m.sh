#!/bin/bash
# round 3, visit m: reference callers (g1), RCCL world-1 test, whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3n; mkdir -p $OUT
ls oracle/_ref/py | head -3
timeout 900 python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_dist.py -q -m gpu 2>&1 | tail -40 > $OUT/pytest_new.log; cat $OUT/pytest_new.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/pytest_all.log; cat $OUT/pytest_all.log
