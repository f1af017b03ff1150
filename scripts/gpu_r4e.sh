#!/bin/bash
# round 4: DCN backward-data restructuring (tables once, early gathers, hoisted dY) -- tests, dcn_r50 A/B, split-K sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r4e}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
D2AMD_DUMP_RATIOS=$OUT/ratios.json timeout 1500 python -m pytest tests/test_gpu_dcn_tc.py tests/test_gpu_dcn_reference.py tests/test_gpu_cshim.py "tests/test_gpu_parity.py" "tests/test_gpu_pooler.py::test_pooler_full_size_per_element_vs_oracle" -m gpu -q -p no:cacheprovider -k "deform or dcn or Deform or pooler_full or cshim or saved or columns" > $OUT/pytest_dcn.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_dcn.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_$name.json 2> $OUT/bench_dcn_$name.err; }
run head X=1
run nohoist D2AMD_DCN_BWD_NO_HOIST=1
run ks28 D2AMD_DCN_BWW_KSPLIT=28
run ks14 D2AMD_DCN_BWW_KSPLIT=14
python - <<PY
import json
for n in ("head","nohoist","ks28","ks14"):
    try:
        d=json.load(open("$OUT/bench_dcn_%s.json"%n)); print(n, d["ms_per_step"], d["roofline"]["kernels_ms"], {k:v["ms_per_step"] for k,v in d["ops"].items()})
    except Exception as e: print(n,"failed",e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dcn -o bench -- python $REPO/bench.py --workload dcn_r50 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_dcn.log 2>&1
cp $(find $OUT/prof_dcn -name "*kernel_stats.csv" | head -1) $OUT/dcn_r50_kernel_stats.csv; rm -rf $OUT/prof_dcn
head -16 $OUT/dcn_r50_kernel_stats.csv | cut -c1-160
