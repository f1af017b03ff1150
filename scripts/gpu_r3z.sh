#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_nms_runs.py tests/test_gpu_rpn.py tests/test_gpu_parity.py -q -p no:cacheprovider -x 2>&1 | tail -1
timeout 120 python scripts/nms_stamps.py 2>&1 | grep "d2amd nms" | cut -c1-900
for rep in 1 2 3; do
D2AMD_LIB_PATH=$PWD/detectron2_amd/lib/libd2amd_prev.so timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'], d['roofline']['kernels_ms']['nms_reduce'])"
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows between', d['ms_per_step'], d['roofline']['kernels_ms']['nms_reduce'])"
done
