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
  timeout 600 python bench.py --steps 200 --warmup 20 --layout $L > $OUT/bench_$L.json 2> $OUT/bench_$L.err; echo "bench $L rc=$?"
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
  echo "== rocprof PMC (HBM traffic), separate passes per counter and op"
  bash $REPO/scripts/gpu_pmc.sh $TAG/pmc nhwc roi_align_box_fwd roi_align_box_bwd roi_align_mask_fwd roi_align_mask_bwd pairwise_iou_rpn
fi
