#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; R=$PWD
cd /tmp
for gy in 0 4 8 16 64; do
  [ $gy = 0 ] && unset D2AMD_NMS_MASK_GY || export D2AMD_NMS_MASK_GY=$gy
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$gy -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pp_$gy/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "nms_mask" in r["Name"]: print("gy=$gy", r["Calls"], round(float(r["AverageNs"])/1e3,2), round(float(r["MinNs"])/1e3,2))
PY
done
