#!/bin/bash
# round 4, second GPU call: the whole suite at HEAD (per-element floors, fp64 ROIAlign, reference tests, DCN reference
# goldens), the default bench line with the two new extra workloads, stand-alone lines of the new workloads with their
# CPU baselines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r4b}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
D2AMD_DUMP_RATIOS=$OUT/ratios_all.json D2AMD_REFERENCE_TEST_REPORT=$OUT/reference_tests_report.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -8 $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --workload maskrcnn_infer > $OUT/bench_maskrcnn_infer.json 2> $OUT/bench_maskrcnn_infer.err; echo "infer rc=$?"; cut -c1-900 $OUT/bench_maskrcnn_infer.json; tail -3 $OUT/bench_maskrcnn_infer.err
timeout 600 python bench.py --workload rrpn_micro > $OUT/bench_rrpn_micro.json 2> $OUT/bench_rrpn_micro.err; echo "rrpn rc=$?"; cut -c1-900 $OUT/bench_rrpn_micro.json; tail -3 $OUT/bench_rrpn_micro.err
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
