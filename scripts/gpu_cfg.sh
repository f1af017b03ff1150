#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/cfg; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider --tb=short -k "pooler or roi_align" 2>&1 | tail -5
cd /tmp
for CFG in "$@"; do
    D2AMD_BWD_CFG=$CFG timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$CFG -o k -- python $REPO/scripts/exp_kernels.py nhwc > $OUT/p.log 2>&1
    echo "cfg=$CFG"; python $REPO/scripts/trace_seq.py $(find $OUT/p_$CFG -name "*kernel_trace.csv") pool_bwd roi_records
    rm -rf $OUT/p_$CFG
done
