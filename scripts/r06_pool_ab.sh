#!/bin/bash
# r06: the K-concatenated tile gather against the per-item pipeline, same box.  The pooler tests on the shipping library,
# then the default bench line per D2AMD_POOL_KCAT setting (profiling build) and the EMPTY-lists ablation.
#   gpurun -- 'bash scripts/r06_pool_ab.sh TAG [notest]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_pool}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
if [ "$2" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_pooler_pair.py -x -q -p no:cacheprovider > $OUT/pytest_pooler.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_pooler.log
fi
line() {  # name, env...
  NAME=$1; shift
  env "$@" D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$NAME.json")); print("$NAME", d["ms_per_step"], d["roofline"]["kernels_ms"])
except Exception as e: print("$NAME failed", e)
PY
}
for REP in 1 2; do
  for K in ${KCATS:-0 1 2 3}; do line kcat${K}_$REP D2AMD_POOL_KCAT=$K; done
done
for K in ${KCATS:-0 1 2 3}; do line kcat${K}_empty D2AMD_POOL_KCAT=$K D2AMD_ABLATE=64; done
for K in ${KCATS:-0 1 2 3}; do
  D2AMD_POOL_KCAT=$K D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so timeout 300 python scripts/pool_bwd_ab.py kcat$K 2>&1 | tail -1
done
