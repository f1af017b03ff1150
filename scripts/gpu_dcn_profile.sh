#!/bin/bash
# dcn_r50 on the box: the bench line (graph replay) and rocprofv3 kernel stats of the same workload.  $1 = tag
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-dcn}; OUT=$REPO/gpurun_out/r05/$TAG; mkdir -p $OUT
timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_r50.json 2> $OUT/bench_dcn_r50.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench_dcn_r50.json")); print("dcn_r50 ms_per_step", d["ms_per_step"], d["roofline"].get("kernels_ms"), d.get("step_tflops"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --workload dcn_r50 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/dcn_r50_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/dcn_r50_kernel_stats.csv")))
for r in rows[:22]:
    print("%-90s calls %6s avg_us %9.2f  pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
