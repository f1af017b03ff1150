#!/bin/bash
# round 4: DCN forward tile heuristic A/B (D2AMD_DCN_FWD_BIG 0 / 1 / 2) + DCN tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r4f}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_dcn_tc.py tests/test_gpu_dcn_reference.py "tests/test_gpu_pooler.py::test_pooler_full_size_per_element_vs_oracle" -m gpu -q -p no:cacheprovider > $OUT/pytest_dcn.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_dcn.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_$name.json 2> $OUT/bench_dcn_$name.err; }
run ws X=1
run binsep D2AMD_DCN_BIN_SEPARATE=1
run ws_again X=2
run ws0_again D2AMD_DCN_BWD_WS0=1
python - <<PY
import json
for n in ("ws","binsep","ws_again","ws0_again"):
    try:
        d=json.load(open("$OUT/bench_dcn_%s.json"%n)); print(n, d["ms_per_step"], d["roofline"]["kernels_ms"], {k:v["ms_per_step"] for k,v in d["ops"].items()})
    except Exception as e: print(n,"failed",e)
PY
