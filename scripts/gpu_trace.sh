#!/bin/bash
# kernel timeline of one replayed step:  gpu_trace.sh TAG [bench flags]
O=gpurun_out/${1:-tr}; mkdir -p $O; shift
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1)
f=$(find /tmp/t -name "*kernel_trace.csv" | head -1)
cp "$f" $O/kernel_trace.csv; python scripts/trace_timeline.py "$f" ${TL_STEP:-2} > $O/timeline.txt; python scripts/trace_gaps.py "$f" > $O/gaps.txt; tail -n 3 $O/timeline.txt
