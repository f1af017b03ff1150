#!/bin/bash
# round 3, visit e: persistent tile workgroups without cross-XCD stealing, padded take counters, level-aware dealing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_graph.py -q -m gpu 2>&1 | tail -4 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 200 python scripts/pool_stamps.py box > $OUT/pool_bwd_box_timeline.txt 2>&1; cat $OUT/pool_bwd_box_timeline.txt; cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_box.pass0
timeout 200 python scripts/pool_stamps.py mask > $OUT/pool_bwd_mask_timeline.txt 2>&1; cat $OUT/pool_bwd_mask_timeline.txt; cp /tmp/pool_stamps.pass0 $OUT/pool_stamps_mask.pass0
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json; d=json.load(open("$OUT/bench_$name.json")); print("$name", d["ms_per_step"], d["roofline"]["kernels_ms"])
PY
}
run dyn_auto A=1
run dyn_deal2 D2AMD_POOL_DEAL=2
run dyn_steal1 D2AMD_POOL_STEAL=1
run dyn_steal7 D2AMD_POOL_STEAL=7
run static_auto D2AMD_POOL_STATIC=1
run static_deal2 D2AMD_POOL_STATIC=1 D2AMD_POOL_DEAL=2
run dyn_auto2 A=1
run static_auto2 D2AMD_POOL_STATIC=1
