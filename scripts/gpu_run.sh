#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (both layouts), per-op microbench, rocprofv3 kernel
# stats and (optionally) PMC traffic passes.  Everything lands in gpurun_out/<tag>/.
#   scripts/gpu_run.sh <tag> [tests|notests] [pmc]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD
TAG=${1:-run}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt
if [ "${2:-tests}" = "tests" ]; then
  echo "== smoke"
  timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
  echo "== pytest gpu"
  timeout 1800 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider --tb=short > $OUT/pytest_all.log 2>&1
  echo "pytest rc=$?"; tail -60 $OUT/pytest_all.log
fi
echo "== bench"
for L in nhwc nchw; do
  timeout 600 python bench.py --steps 30 --warmup 5 --layout $L > $OUT/bench_$L.json 2> $OUT/bench_$L.err; echo "bench $L rc=$?"
  cat $OUT/bench_$L.json; tail -3 $OUT/bench_$L.err
done
echo "== microbench"
for L in nhwc nchw; do
  timeout 900 python scripts/microbench.py $L > $OUT/micro_$L.json 2> $OUT/micro_$L.err; echo "micro $L rc=$?"
  cat $OUT/micro_$L.json; tail -3 $OUT/micro_$L.err
done
echo "== rocprof kernel stats"
cd /tmp
for L in nhwc; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$L -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --layout $L --no-cpu-baseline > $OUT/prof_$L.log 2>&1; echo "rocprof $L rc=$?"
  f=$(find $OUT/prof_$L -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f | cut -c1-220
  find $OUT/prof_$L -type f -name "*kernel_trace.csv" -size +8M -delete
done
if [ "${3:-}" = "pmc" ]; then
  echo "== rocprof PMC (HBM traffic), separate passes"
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_$CNT -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --layout nhwc --no-cpu-baseline > $OUT/pmc_$CNT.log 2>&1; echo "pmc $CNT rc=$?"
    python $REPO/scripts/pmc_summary.py $OUT/pmc_$CNT $CNT > $OUT/pmc_$CNT.summary.json 2>> $OUT/pmc_$CNT.log; cat $OUT/pmc_$CNT.summary.json | cut -c1-1500
    find $OUT/pmc_$CNT -type f -size +4M -delete
  done
fi
