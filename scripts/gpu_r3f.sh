#!/bin/bash
# round 3, visit f: split tile lists in the pooler backward -- parity, timeline, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 200 python scripts/pool_stamps.py box > $OUT/pool_bwd_box_timeline.txt 2>&1; cat $OUT/pool_bwd_box_timeline.txt; cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_box.pass0
timeout 200 python scripts/pool_stamps.py mask > $OUT/pool_bwd_mask_timeline.txt 2>&1; cat $OUT/pool_bwd_mask_timeline.txt; cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_mask.pass0
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json; d=json.load(open("$OUT/bench_$name.json")); print("$name", d["ms_per_step"], d["roofline"]["kernels_ms"])
PY
}
run split A=1
run nosplit D2AMD_POOL_NOSPLIT=1


run split2 A=1
run nosplit2 D2AMD_POOL_NOSPLIT=1
