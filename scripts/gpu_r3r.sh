#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_connected_step.py -q -m gpu 2>&1 | tail -3
timeout 200 python scripts/pool_stamps.py box > $OUT/pool_bwd_box_timeline.txt 2>&1; cat $OUT/pool_bwd_box_timeline.txt
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernels_ms'])"; }
run prefetch A=1
run noprefetch D2AMD_ABLATE=16
run prefetch2 A=1
run noprefetch2 D2AMD_ABLATE=16
