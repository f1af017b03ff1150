#!/bin/bash
# The round's evidence set in ONE gpurun call (~4 min of box time): smoke, the whole GPU suite (per-element margins and
# the list of executed reference tests dumped), every bench.py workload as its own line with its CPU baseline, the
# default line (what the driver runs), rocprofv3 kernel stats of the three heaviest workloads.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_evidence.sh TAG'   -> gpurun_out/TAG/  (copy what is kept to profiles/rNN/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-evidence}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; git -C $REPO rev-parse HEAD >> $OUT/gpu.txt 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
D2AMD_DUMP_RATIOS=$OUT/parity_margins.json D2AMD_REFERENCE_TEST_REPORT=$OUT/reference_tests_report.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for WL in maskrcnn_infer rrpn_micro retinanet_100k dcn_r50; do
  timeout 600 python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; echo "$WL rc=$?"
done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"; cut -c1-260 $OUT/bench_default.json
python - <<PY
import json
for n in ("default","maskrcnn_infer","rrpn_micro","retinanet_100k","dcn_r50"):
    try:
        d=json.load(open("$OUT/bench_%s.json"%n)); print(n, d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("kernels_ms"))
    except Exception as e: print(n,"failed",e)
PY
cd /tmp
for WL in maskrcnn_train dcn_r50 maskrcnn_infer retinanet_100k rrpn_micro; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -o bench -- python $REPO/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof_$WL.log 2>&1
  cp $(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1) $OUT/${WL}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof_$WL
done
