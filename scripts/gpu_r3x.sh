#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3x; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dcn_tc.py tests/test_gpu_parity.py -q -m gpu -k "deform or dcn or channels_last or full_size or Deform" 2>&1 | tail -5
for L in nhwc nchw nhwc nchw; do
  timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline --layout $L > $OUT/bench_dcn_$L.json 2>$OUT/err_$L.txt; python -c "import json; d=json.load(open('$OUT/bench_dcn_$L.json')); print('$L', d['ms_per_step'], d['roofline']['kernels_ms'], d['roofline'].get('gather_floor',{}).get('frac_of_floor'))"
done
