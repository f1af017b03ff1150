#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${TAG:-r3q}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_connected_step.py tests/test_gpu_graph.py tests/test_gpu_mask_head.py tests/test_gpu_rpn.py -q -m gpu 2>&1 | tail -4
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernels_ms'])"; }
run side1 A=1
run side0 D2AMD_SIDE_BINNING=0
run side1b A=1
run side0b D2AMD_SIDE_BINNING=0
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
