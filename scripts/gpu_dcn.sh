#!/bin/bash
# GPU visit for the DCN MFMA path: parity tests, config sweep, rocprofv3 kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-dcn}
mkdir -p $OUT
echo "== pytest dcn tc"
timeout 900 python -m pytest tests/test_gpu_dcn_tc.py -q --timeout=300 -p no:cacheprovider --tb=short -x > $OUT/pytest_tc.log 2>&1
echo "rc=$?"; tail -40 $OUT/pytest_tc.log
echo "== pytest old dcn"
timeout 600 python -m pytest tests/test_gpu_parity.py -q --timeout=300 -p no:cacheprovider --tb=short -k "deform" > $OUT/pytest_dcn.log 2>&1
echo "rc=$?"; tail -15 $OUT/pytest_dcn.log
echo "== bench default"
timeout 300 python scripts/dcn_bench.py > $OUT/dcn_default.json 2> $OUT/dcn_default.err; echo "rc=$?"; cat $OUT/dcn_default.json; tail -3 $OUT/dcn_default.err
echo "== bench cfg sweep (fwd)"
timeout 300 python scripts/dcn_bench.py --fwd-only --cfgs "4,1,4,1;4,1,2,1;4,1,1,1;2,1,4,1;2,2,2,1;4,2,2,1;4,2,1,1;4,2,2,2;4,2,1,2;4,2,1,3;4,1,2,2" > $OUT/dcn_sweep.json 2> $OUT/dcn_sweep.err; echo "rc=$?"; cat $OUT/dcn_sweep.json; tail -3 $OUT/dcn_sweep.err
echo "== patch sweep (bwd)"
for R in -1 2 4; do D2AMD_DCN_PATCH_R=$R timeout 300 python scripts/dcn_bench.py > $OUT/dcn_patch_$R.json 2>> $OUT/dcn_sweep.err; echo "R=$R"; cat $OUT/dcn_patch_$R.json; done
echo "== rocprof"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o dcn -- python $REPO/scripts/dcn_bench.py > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f | cut -c1-200
find $OUT/prof -type f -name "*kernel_trace.csv" -size +8M -delete
