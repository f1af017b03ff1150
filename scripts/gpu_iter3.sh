#!/bin/bash
# one iteration on the GPU box: the selection / NMS tests + the bench lines they move
O=gpurun_out/${1:-it3}; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nms_runs.py tests/test_gpu_rpn.py tests/test_gpu_dense.py -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$?; cut -c1-330 $O/bench.json
timeout 200 python bench.py --no-cpu-baseline --workload retinanet_100k > $O/bench_ret.json 2> $O/bench_ret.err; echo rc=$?; cut -c1-330 $O/bench_ret.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload retinanet_100k > /dev/null 2>&1)
f=$(find /tmp/p -name "*kernel_stats.csv" | head -1); cp "$f" $O/retinanet_100k_kernel_stats.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap > /dev/null 2>&1)
f=$(find /tmp/q -name "*kernel_stats.csv" | head -1); cp "$f" $O/maskrcnn_train_kernel_stats.csv
