#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (both layouts), rocprofv3 kernel stats.
# Everything lands in gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
OUT=gpurun_out/run1
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python __graft_entry__.py --smoke >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest_x.log 2>&1; echo "pytest -x rc=$?" | tee -a $OUT/pytest_x.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider --tb=short > $OUT/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/pytest_all.log
tail -40 $OUT/pytest_all.log
echo "== bench"
for L in nhwc nchw; do
  timeout 600 python bench.py --steps 30 --warmup 5 --layout $L > $OUT/bench_$L.json 2> $OUT/bench_$L.err; echo "bench $L rc=$?"
  cat $OUT/bench_$L.json
done
echo "== rocprof"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_nhwc -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 3 --layout nhwc --no-cpu-baseline > $OLDPWD/$OUT/prof_nhwc.log 2>&1; echo "rocprof rc=$?"
cd $OLDPWD
find $OUT/prof_nhwc -name "*stats*" | head; 
for f in $(find $OUT/prof_nhwc -name "*kernel_stats*.csv" | head -1); do head -30 $f; done
# keep only the small summaries (traces can be large)
find $OUT/prof_nhwc -type f ! -name "*stats*" -size +2M -delete
