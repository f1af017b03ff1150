#!/bin/bash
# rocprofv3 kernel stats of the three bench workloads -> gpurun_out/$1/
O=gpurun_out/${1:-prof3}; mkdir -p $O
export TMPDIR=/tmp
for wl in maskrcnn_train retinanet_100k dcn_r50; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); echo "== $wl $f"; head -30 "$f" | cut -c1-220
  rm -f $(find $O/prof_$wl -name "*kernel_trace.csv")
done
