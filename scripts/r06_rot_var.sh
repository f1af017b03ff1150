#!/bin/bash
# r06: same-box sweep of variant builds of the rotated NMS mask tile on rrpn_micro:  bash scripts/r06_rot_var.sh TAG lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_rot_var}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; shift
for REP in 1 2; do for LIB in "$@"; do
  D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/$LIB timeout 300 python bench.py --workload rrpn_micro --no-cpu-baseline > $OUT/bench_${LIB}_$REP.json 2> $OUT/bench_${LIB}_$REP.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${LIB}_$REP.json")); print("$LIB", $REP, d["ms_per_step"], d.get("roofline", {}).get("kernels_ms"))
except Exception as e: print("$LIB failed", e)
PY
done; done
