#!/bin/bash
# A/B of the dcn_r50 workload under environment switches, on ONE box (boxes differ by several percent):
#   gpurun -- 'bash scripts/dcn_ab.sh TAG name1:VAR=VALUE name2:VAR=VALUE ...'   ("name:X=1" = the defaults)
# Runs the DCN test files first, then one bench line per variant -> gpurun_out/TAG/bench_dcn_<name>.json + a summary.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-dcn_ab}; shift; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_dcn_tc.py tests/test_gpu_dcn_reference.py -m gpu -q -p no:cacheprovider > $OUT/pytest_dcn.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_dcn.log
NAMES=""
for V in "$@"; do
  name=${V%%:*}; kv=${V#*:}; NAMES="$NAMES $name"
  env "$kv" timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_$name.json 2> $OUT/bench_dcn_$name.err
done
python - $NAMES <<PY
import json, sys
for n in sys.argv[1:]:
    try:
        d=json.load(open("$OUT/bench_dcn_%s.json"%n)); print(n, d["ms_per_step"], d["roofline"]["kernels_ms"], {k:v["ms_per_step"] for k,v in d["ops"].items()})
    except Exception as e: print(n,"failed",e)
PY
