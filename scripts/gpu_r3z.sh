#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_label_sample.py tests/test_gpu_connected_step.py -q -p no:cacheprovider -x 2>&1 | tail -1
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('keys on the side branch', d['ms_per_step'])"
done
