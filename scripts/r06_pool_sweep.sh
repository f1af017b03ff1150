#!/bin/bash
# r06: default bench line per environment setting (profiling build):  bash scripts/r06_pool_sweep.sh TAG "name:VAR=V,VAR2=V2" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_sweep}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; shift
for REP in 1 2; do for S in "$@"; do
  NAME=${S%%:*}; ENVS=$(echo "${S#*:}" | tr ',' ' ')
  env $ENVS D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads ${BENCH_ARGS} > $OUT/bench_${NAME}_$REP.json 2> $OUT/bench_${NAME}_$REP.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${NAME}_$REP.json")); print("$NAME", $REP, d["ms_per_step"], d["roofline"]["kernels_ms"])
except Exception as e: print("$NAME failed", e)
PY
done; done
