#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; R=$PWD
PREV=$PWD/detectron2_amd/lib/libd2amd_prev.so
timeout 600 python -m pytest tests/test_gpu_label_sample.py tests/test_gpu_connected_step.py -q -p no:cacheprovider -x 2>&1 | tail -1
cd /tmp
for v in prev new; do
  [ $v = prev ] && export D2AMD_LIB_PATH=$PREV || unset D2AMD_LIB_PATH
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$v -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pp_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "label_sample" in r["Name"]: print("$v", r["Calls"], round(float(r["AverageNs"])/1e3,2), round(float(r["MinNs"])/1e3,2), r["Name"][:40])
PY
done
