#!/bin/bash
# r06: same-box A/B of the DCN backward on bench.py --workload dcn_r50 with profiling builds:
#   bash scripts/r06_dcn_ab.sh TAG name[:VAR=V,...] ...      (lib/libd2amd_prof.so; VARs are d2_prof_env switches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_dcn_ab}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; shift
for REP in 1 2 3; do for S in "$@"; do
  NAME=$(echo $S | cut -d: -f1); ENVS=$(echo $S | cut -s -d: -f2 | tr ',' ' ')
  env $ENVS D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_${NAME}_$REP.json 2> $OUT/bench_${NAME}_$REP.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${NAME}_$REP.json")); print("$NAME", $REP, d["ms_per_step"], d.get("roofline", {}).get("kernels_ms"))
except Exception as e: print("$NAME failed", e)
PY
done; done
