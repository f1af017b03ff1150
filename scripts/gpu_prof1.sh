#!/bin/bash
# rocprofv3 kernel stats of one bench workload -> gpurun_out/$1/ ; $2 = workload, rest = extra bench args
O=gpurun_out/${1:-prof1}; wl=${2:-maskrcnn_train}; shift; shift
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1)
f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); echo "== $wl $f"; head -${LINES_SHOW:-40} "$f" | cut -c1-200
rm -f $(find $O/prof_$wl -name "*kernel_trace.csv")
