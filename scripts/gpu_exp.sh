#!/bin/bash
# quick experiment: optional pytest filter, then kernel stats of scripts/exp_kernels.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; TAG=${1:-exp}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
if [ -n "${2:-}" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider --tb=short -k "$2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/scripts/exp_kernels.py ${3:-nhwc} > $OUT/prof.log 2>&1; echo "rocprof rc=$?"; tail -3 $OUT/prof.log
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:110]}")
PY
find $OUT/prof -type f -name "*kernel_trace.csv" -size +8M -delete
