#!/bin/bash
O=gpurun_out/${1:-it}; mkdir -p $O
(time python -m pytest tests/test_gpu_matcher.py tests/test_gpu_nms_runs.py tests/test_gpu_rpn.py tests/test_gpu_graph.py tests/test_gpu_pooler.py -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$?; cut -c1-330 $O/bench.json
python bench.py --no-cpu-baseline --no-overlap > $O/bench_noov.json 2> $O/bench_noov.err; echo rc=$?; cut -c1-330 $O/bench_noov.json
python bench.py --no-cpu-baseline --no-graph > $O/bench_eager.json 2> $O/bench_eager.err; echo rc=$?; cut -c1-330 $O/bench_eager.json
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap > /dev/null 2>&1)
f=$(find /tmp/p -name "*kernel_stats.csv" | head -1); cp "$f" $O/maskrcnn_train_kernel_stats.csv
