#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for rep in 1 2; do
D2AMD_POOL_STRIDE=0 timeout 120 python scripts/pool_bwd_ab.py takes 2>&1 | tail -1
done
for k in 0 512; do
echo "== takes, stamp2 at $k"; D2AMD_ABLATE=$k D2AMD_POOL_STRIDE=0 timeout 120 python scripts/pool_stamps.py box 2>&1 | grep "rois  :\|write :\|scan  :\|per work"
done
D2AMD_POOL_STRIDE=0 timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('takes', d['ms_per_step'], d['roofline']['kernels_ms'])"
