#!/bin/bash
# round 2: full GPU suite, the three bench workloads, rocprofv3 kernel stats of each
O=gpurun_out/r2c; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log
for wl in maskrcnn_train retinanet_100k dcn_r50; do
  timeout 600 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "rc=$? $wl"; head -c 3000 $O/bench_$wl.json; echo; tail -3 $O/bench_$wl.err
done
export TMPDIR=/tmp
for wl in maskrcnn_train retinanet_100k dcn_r50; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); echo "== $wl $f"; head -25 "$f" | cut -c1-200
done
