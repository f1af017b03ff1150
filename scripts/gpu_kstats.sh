#!/bin/bash
# rocprofv3 kernel stats + one replayed step's timeline of a bench workload.  $1 = workload, $2 = tag
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; WL=$1; TAG=${2:-$1}; OUT=$REPO/gpurun_out/${ROUND:-r06}/$TAG; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads ${@:3} > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/${WL}_kernel_stats.csv 2>/dev/null
cp $(find $OUT/prof -name "*kernel_trace.csv" | head -1) /tmp/ktrace.csv 2>/dev/null; rm -rf $OUT/prof
python - <<PY
import csv, re
rows = list(csv.DictReader(open("/tmp/ktrace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# steps = runs of kernels separated by > 12 us of idle; print the most common step shape's LAST instance
steps, cur, end = [], [ts[0]], ts[0][1]
for b in ts[1:]:
    if b[0] - end > 12000: steps.append(cur); cur = [b]; end = b[1]
    else: cur.append(b); end = max(end, b[1])
steps.append(cur)
from collections import Counter
common = Counter(len(s) for s in steps if len(s) > 10).most_common(1)[0][0]
st = [s for s in steps if len(s) == common][-1]
t0 = st[0][0]
out = []
for s, e, n in st:
    n = re.sub(r"^void |d2amd::|at::native::|\\(.*", "", n)[:72]
    out.append("%8.1f %8.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
open("$OUT/${WL}_step_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:150])); print("kernels in step:", len(st), "span us:", (max(x[1] for x in st) - t0) / 1e3, "steps of that shape:", sum(1 for s in steps if len(s) == common))
PY
