#!/bin/bash
# rocprofv3 kernel stats (csv) of bench.py for both layouts + isolated per-op timings.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD
OUT=$PWD/gpurun_out/${1:-prof}
mkdir -p $OUT
for L in nhwc nchw; do
  timeout 300 python scripts/microbench.py $L > $OUT/micro_$L.json 2> $OUT/micro_$L.err; echo "micro $L rc=$?"; cat $OUT/micro_$L.json
done
cd /tmp
for L in nhwc nchw; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$L -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --layout $L --no-cpu-baseline > $OUT/prof_$L.log 2>&1; echo "rocprof $L rc=$?"
  f=$(find $OUT/prof_$L -name "*kernel_stats.csv" | head -1); echo $f; head -25 $f
  find $OUT/prof_$L -type f -name "*kernel_trace.csv" -size +8M -delete
done
