#!/bin/bash
# round 3, visit h: connected one-graph step -- tests, bench (connected | disconnected), kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_connected_step.py tests/test_gpu_label_sample.py tests/test_gpu_rpn.py -q -m gpu 2>&1 | tail -60 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_connected.json 2> $OUT/bench_connected.err; tail -5 $OUT/bench_connected.err; cut -c1-3000 $OUT/bench_connected.json
timeout 300 python bench.py --no-cpu-baseline --disconnected --no-extra-workloads > $OUT/bench_disconnected.json 2> $OUT/bench_disconnected.err; tail -3 $OUT/bench_disconnected.err
python - <<PY
import json
for n in ("connected","disconnected"):
    try:
        d=json.load(open("$OUT/bench_%s.json"%n)); print(n, d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["ops"].items()})
        if "extra_workloads" in d: print({k:(v["ms_per_step"],v["value"]) for k,v in d["extra_workloads"].items()})
    except Exception as e: print(n, "failed", e)
PY
