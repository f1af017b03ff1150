#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/ablate; mkdir -p $OUT; rm -rf $OUT/prof
export PLAN_OUT=$OUT/plan.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o ab -- python $REPO/scripts/dcn_ablate.py "$@" > $OUT/prof.log 2>&1; echo "rocprof rc=$?"; tail -3 $OUT/prof.log
python $REPO/scripts/dcn_ablate_parse.py $OUT/plan.json $(find $OUT/prof -name "*kernel_trace.csv" | head -1)
rm -rf $OUT/prof
