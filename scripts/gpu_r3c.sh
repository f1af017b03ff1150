#!/bin/bash
# round 3, visit c: subsample tests, per-workgroup timeline of the CURRENT pooler backward, baseline bench + kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3c; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_subsample.py -q -m gpu 2>&1 | tail -5 > $OUT/pytest_subsample.log; cat $OUT/pytest_subsample.log
timeout 200 python scripts/pool_stamps.py box > $OUT/pool_bwd_box_timeline.txt 2>&1; cat $OUT/pool_bwd_box_timeline.txt
cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_box.pass0 2>/dev/null
timeout 200 python scripts/pool_stamps.py mask > $OUT/pool_bwd_mask_timeline.txt 2>&1; cat $OUT/pool_bwd_mask_timeline.txt
cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_mask.pass0 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-1500
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; head -40 $f | cut -c1-200
find $OUT/prof -type f -name "*kernel_trace.csv" -size +4M -delete
