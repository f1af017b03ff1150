#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
for rep in 1 2 3; do
D2AMD_BENCH_FORK=nms timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork=nms', d['ms_per_step'])"
D2AMD_BENCH_FORK=start timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork=start side first', d['ms_per_step'])"
D2AMD_BENCH_FORK=start D2AMD_BENCH_FORK_ORDER=main timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork=start main first', d['ms_per_step'])"
done
