#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
for t in 512 1024; do
D2AMD_LIB_PATH=$PWD/detectron2_amd/lib/libd2amd_t$t.so timeout 600 python -m pytest tests/test_gpu_rpn.py tests/test_gpu_subsample.py -q -p no:cacheprovider -x 2>&1 | tail -1
done
for rep in 1 2 3; do
for t in 256 512 1024; do
LIBP=$PWD/detectron2_amd/lib/libd2amd_t$t.so; [ $t = 256 ] && LIBP=$PWD/detectron2_amd/lib/libd2amd.so
D2AMD_LIB_PATH=$LIBP timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank threads $t', d['ms_per_step'])"
done; done
