#!/bin/bash
# A/B on one box: where the binning of the pooler backward runs (D2AMD_PREBIN), the counting sort of the large NMS
O=gpurun_out/${1:-it4}; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nms_runs.py tests/test_gpu_dense.py -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
(time D2AMD_PREBIN=side timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py -x -q) > $O/pytest_side.log 2>&1; tail -3 $O/pytest_side.log
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err; echo rc=$?; cut -c1-330 $O/bench$i.json | grep -o '"ms_per_step": [0-9.]*'
D2AMD_PREBIN=side timeout 200 python bench.py --no-cpu-baseline > $O/bench_side$i.json 2> $O/bench_side$i.err; echo rc=$?; cut -c1-330 $O/bench_side$i.json | grep -o '"ms_per_step": [0-9.]*'
done
timeout 200 python bench.py --no-cpu-baseline --workload retinanet_100k > $O/bench_ret.json 2> $O/bench_ret.err; echo rc=$?; cut -c1-330 $O/bench_ret.json | grep -o '"ms_per_step": [0-9.]*'
