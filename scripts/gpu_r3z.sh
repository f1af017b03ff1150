#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms'])"
timeout 300 python bench.py --workload retinanet_100k --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet', d['ms_per_step'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3zo -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > /dev/null 2>&1
