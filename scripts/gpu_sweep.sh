#!/bin/bash
# scripts/gpu_sweep.sh <fwd|bwd> <ENV> "<v1;v2>" [shapes]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/sweep; mkdir -p $OUT; rm -rf $OUT/prof
export PLAN_OUT=$OUT/plan.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o ab -- python $REPO/scripts/dcn_sweep.py "$@" > $OUT/prof.log 2>&1; echo "rocprof rc=$?"; grep -i "error\|Traceback" -A5 $OUT/prof.log | head -20
python $REPO/scripts/dcn_ablate_parse.py $OUT/plan.json $(find $OUT/prof -name "*kernel_trace.csv" | head -1)
rm -rf $OUT/prof
