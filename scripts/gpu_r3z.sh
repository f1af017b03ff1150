#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py tests/test_gpu_connected_step.py tests/test_gpu_reference_callers.py -q -p no:cacheprovider -x 2>&1 | tail -2
for rep in 1 2 3; do
D2AMD_POOL_NOFUSEDBIN=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two launches', d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused binning', d['ms_per_step'])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3zf -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > /dev/null 2>&1
grep -i "tile_lists\|roi_records\|fill\|memset" $(find $GRAFT_REPO_ROOT/gpurun_out/r3zf -name "*kernel_stats.csv") | cut -c1-160
