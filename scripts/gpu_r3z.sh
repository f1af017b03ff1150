#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_dcn_tc.py tests/test_gpu_cshim.py tests/test_gpu_parity.py -q -p no:cacheprovider -x 2>&1 | tail -1
for rep in 1; do
D2AMD_LIB_PATH=$PWD/detectron2_amd/lib/libd2amd_prev.so timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'])"
timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merged zero/cvt', d['ms_per_step'])"
done
