#!/bin/bash
# dcn_r50 on the box: the bench line (graph replay) and rocprofv3 kernel stats of the same workload.  $1 = tag
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-dcn}; OUT=$REPO/gpurun_out/${ROUND:-r06}/$TAG; mkdir -p $OUT
timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_r50.json 2> $OUT/bench_dcn_r50.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench_dcn_r50.json")); print("dcn_r50 ms_per_step", d["ms_per_step"], d["roofline"].get("kernels_ms"), d.get("step_tflops"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --workload dcn_r50 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/dcn_r50_kernel_stats.csv 2>/dev/null
cp $(find $OUT/prof -name "*kernel_trace.csv" | head -1) /tmp/dcn_r50_kernel_trace.csv 2>/dev/null; rm -rf $OUT/prof
python - <<PY2
import csv, collections, re
rows = list(csv.DictReader(open("/tmp/dcn_r50_kernel_trace.csv")))
g = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"^void d2amd::|^d2amd::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)[:60]
    g[(n, int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for (n, grid), v in sorted(g.items()):
    if len(v) < 20: continue
    v.sort()
    out.append("%-62s grid %8d calls %5d  median %8.2f us  p10 %8.2f" % (n, grid, len(v), v[len(v) // 2], v[len(v) // 10]))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(len(rows) / 41 + 1):]  # ~ the last replayed step (41 = warmup + timed + capture passes of this command)
t0 = int(last[0]["Start_Timestamp"])
with open("$OUT/dcn_r50_last_step_timeline.txt", "w") as f:
    for r in last:
        n = re.sub(r"^void d2amd::|^d2amd::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:56]
        f.write("%9.2f %8.2f  q%-3s %-58s grid %8d\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r.get("Queue_Id", "?"), n, int(r["Grid_Size_X"])))
open("$OUT/dcn_r50_kernels_by_shape.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY2
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/dcn_r50_kernel_stats.csv")))
for r in rows[:22]:
    print("%-90s calls %6s avg_us %9.2f  pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
