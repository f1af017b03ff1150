#!/bin/bash
# r06: same-box A/B of library builds on bench.py --workload maskrcnn_infer:  bash scripts/r06_infer_ab.sh TAG name:lib.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_infer_ab}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; shift
for REP in 1 2 3; do for S in "$@"; do
  NAME=$(echo $S | cut -d: -f1); LIB=$(echo $S | cut -d: -f2)
  D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/$LIB timeout 300 python bench.py --workload maskrcnn_infer --no-cpu-baseline > $OUT/bench_${NAME}_$REP.json 2> $OUT/bench_${NAME}_$REP.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${NAME}_$REP.json")); print("$NAME", $REP, d["ms_per_step"], d.get("roofline", {}).get("kernels_ms"))
except Exception as e: print("$NAME failed", e)
PY
done; done
