#!/bin/bash
# round 4, third GPU call: the saved-column weight gradient (DCN tests + reference-kernel comparisons), dcn_r50 A/B,
# the pooler full-size test, the two new bench workloads.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r4c}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
D2AMD_DUMP_RATIOS=$OUT/ratios.json timeout 1500 python -m pytest tests/test_gpu_dcn_tc.py tests/test_gpu_dcn_reference.py tests/test_gpu_parity.py tests/test_gpu_cshim.py "tests/test_gpu_pooler.py::test_pooler_full_size_per_element_vs_oracle" -m gpu -q -p no:cacheprovider > $OUT/pytest_dcn.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_dcn.log
timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_saved.json 2> $OUT/bench_dcn_saved.err; echo "dcn rc=$?"; tail -2 $OUT/bench_dcn_saved.err
D2AMD_DCN_NO_SAVED_COL=1 timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_regather.json 2> /dev/null
python - <<PY
import json
for n in ("saved","regather"):
    try:
        d=json.load(open("$OUT/bench_dcn_%s.json"%n)); print(n, d["ms_per_step"], d["roofline"]["kernels_ms"], {k:v["ms_per_step"] for k,v in d["ops"].items()})
    except Exception as e: print(n,"failed",e)
PY
timeout 600 python bench.py --workload rrpn_micro > $OUT/bench_rrpn_micro.json 2> $OUT/bench_rrpn_micro.err; echo "rrpn rc=$?"; cut -c1-1200 $OUT/bench_rrpn_micro.json; tail -3 $OUT/bench_rrpn_micro.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dcn -o bench -- python $REPO/bench.py --workload dcn_r50 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_dcn.log 2>&1
cp $(find $OUT/prof_dcn -name "*kernel_stats.csv" | head -1) $OUT/dcn_r50_kernel_stats.csv; rm -rf $OUT/prof_dcn
head -30 $OUT/dcn_r50_kernel_stats.csv | cut -c1-200
