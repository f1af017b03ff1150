#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3s; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernels_ms'])"; }
run base A=1
run noorder D2AMD_FWD_NO_ORDER=1
run prebin_all D2AMD_PREBIN=all
run prebin_chained D2AMD_PREBIN=chained
run base2 A=1
run noorder2 D2AMD_FWD_NO_ORDER=1
env A=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --no-overlap > $OUT/bench_nooverlap.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/bench_nooverlap.json')); print('nooverlap', d['ms_per_step'])"
