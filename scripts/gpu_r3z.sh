#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; R=$PWD
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py tests/test_gpu_connected_step.py tests/test_gpu_reference_callers.py -q -p no:cacheprovider -x 2>&1 | tail -1
for rep in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planner on demand', d['ms_per_step'], d['roofline']['kernels_ms'])"; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/scripts/pool_bwd_ab.py v > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tile_lists" in r["Name"] or "pool_bwd_mfma" in r["Name"]: print(r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Name"][:60])
PY
