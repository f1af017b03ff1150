#!/bin/bash
O=gpurun_out/${1:-it1}; mkdir -p $O
(time python -m pytest tests/test_gpu_nms_runs.py tests/test_gpu_rpn.py tests/test_gpu_dense.py tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_cshim.py -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python scripts/glue_trace.py > $O/glue.txt 2>&1; tail -45 $O/glue.txt
for wl in maskrcnn_train retinanet_100k; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "rc=$? $wl"; cut -c1-300 $O/bench_$wl.json
done
export TMPDIR=/tmp
for wl in maskrcnn_train retinanet_100k; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); cp "$f" $O/${wl}_kernel_stats.csv
  rm -rf $O/prof_$wl
done
