#!/bin/bash
# rrpn_micro with the fused rotated pooler and level by level (D2AMD_ROT_POOLER_LOOP=1), same box, + the pooler's tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; OUT=gpurun_out/${1:-rrpn_ab}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pooler_rotated.py tests/test_gpu_pooler.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for R in 1 2; do for M in 0 1; do
  D2AMD_ROT_POOLER_LOOP=$M timeout 300 python bench.py --workload rrpn_micro --no-cpu-baseline > $OUT/rrpn_loop${M}_$R.json 2> $OUT/rrpn_loop${M}_$R.err
  python -c "import json; d=json.load(open('$OUT/rrpn_loop${M}_$R.json')); print('loop=$M', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['ops'].items()}, d['roofline']['kernels_ms'])"
done; done
