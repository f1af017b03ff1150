#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_subsample.py tests/test_gpu_rpn.py tests/test_gpu_connected_step.py tests/test_gpu_graph.py tests/test_gpu_dense.py tests/test_gpu_reference_callers.py -q -m gpu 2>&1 | tail -3
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernels_ms'])"; }
run fused A=1

run fused2 A=1

