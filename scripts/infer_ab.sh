#!/bin/bash
# maskrcnn_infer with the fused box-head inference and with the reference's per-image data flow (same box) + its tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; OUT=gpurun_out/${1:-infer_ab}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fast_rcnn.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for R in 1 2; do for M in 0 1; do
  D2AMD_BENCH_INFER_LOOP=$M timeout 300 python bench.py --workload maskrcnn_infer --no-cpu-baseline > $OUT/infer_loop${M}_$R.json 2> $OUT/infer_loop${M}_$R.err
  python -c "import json; d=json.load(open('$OUT/infer_loop${M}_$R.json')); print('loop=$M', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['ops'].items()}, d['config']['candidates_above_score_thresh'], d['config']['detections'])"
done; done
