#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
PREV=$PWD/detectron2_amd/lib/libd2amd_prev.so
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py tests/test_gpu_connected_step.py tests/test_gpu_reference_callers.py -q -p no:cacheprovider -x 2>&1 | tail -2
for rep in 1 2; do
D2AMD_LIB_PATH=$PREV timeout 120 python scripts/pool_bwd_ab.py prev 2>&1 | tail -1
timeout 120 python scripts/pool_bwd_ab.py sorted 2>&1 | tail -1
done
echo "== prev"; D2AMD_LIB_PATH=$PREV timeout 120 python scripts/pool_stamps.py box 2>&1 | grep "total :\|per work\|start p50"
echo "== sorted"; timeout 120 python scripts/pool_stamps.py box 2>&1 | grep "total :\|per work\|start p50"
echo "== sorted mask"; timeout 120 python scripts/pool_stamps.py mask 2>&1 | grep "total :\|per work\|start p50"
for rep in 1 2; do
D2AMD_LIB_PATH=$PREV timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'], d['roofline']['kernels_ms'])"
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sorted', d['ms_per_step'], d['roofline']['kernels_ms'])"
done
