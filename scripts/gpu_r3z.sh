#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_dcn_tc.py -q -p no:cacheprovider -x 2>&1 | tail -1
timeout 120 python scripts/dcn_bww_ab.py coop2 2>&1 | tail -1
timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cooperative', d['ms_per_step'], d['roofline']['kernels_ms'])"
