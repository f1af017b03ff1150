#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${TAG:-r3q}; mkdir -p $OUT

run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernels_ms'])"; }




cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
