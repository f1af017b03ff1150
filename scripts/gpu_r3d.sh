#!/bin/bash
# round 3, visit d: persistent tile workgroups of the pooler backward -- parity, timeline, A/B against one workgroup per slot
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3d; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py tests/test_gpu_subsample.py -q -m gpu 2>&1 | tail -8 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 200 python scripts/pool_stamps.py box > $OUT/pool_bwd_box_timeline.txt 2>&1; cat $OUT/pool_bwd_box_timeline.txt; cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_box.pass0
timeout 200 python scripts/pool_stamps.py mask > $OUT/pool_bwd_mask_timeline.txt 2>&1; cat $OUT/pool_bwd_mask_timeline.txt; cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_mask.pass0
for v in dynamic static dynamic static; do
  if [ $v = static ]; then export D2AMD_POOL_STATIC=1; else unset D2AMD_POOL_STATIC; fi
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json; d=json.load(open("$OUT/bench_$v.json")); print("$v", d["ms_per_step"], d["roofline"]["kernels_ms"])
PY
done
unset D2AMD_POOL_STATIC
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; head -8 $f | cut -c1-160
find $OUT/prof -type f -name "*kernel_trace.csv" -size +4M -delete
