#!/bin/bash
# round 4, first GPU call: goldens from the reference's own DCN kernels, the new parity tests (DCN vs compiled reference,
# the reference's own unit tests on this surface, full-size pooler per-element check), the whole suite with the
# per-element fp32 bounds, and the default bench line at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r4a}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt; nproc >> $OUT/gpu.txt
timeout 600 python tests/golden/make_dcn_reference_gpu.py $OUT/dcn_reference_gpu.npz > $OUT/make_golden.log 2>&1; echo "golden rc=$?"; tail -2 $OUT/make_golden.log
cp $OUT/dcn_reference_gpu.npz tests/golden/dcn_reference_gpu.npz
D2AMD_DUMP_RATIOS=$OUT/ratios_new.json D2AMD_REFERENCE_TEST_REPORT=$OUT/reference_tests_report.txt timeout 1200 python -m pytest tests/test_gpu_dcn_reference.py tests/test_gpu_reference_tests.py "tests/test_gpu_pooler.py::test_pooler_full_size_per_element_vs_oracle" -m gpu -q -p no:cacheprovider > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -5 $OUT/pytest_new.log
D2AMD_DUMP_RATIOS=$OUT/ratios_all.json timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_dcn_reference.py --deselect tests/test_gpu_reference_tests.py > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_default.json
