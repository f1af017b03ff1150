#!/bin/bash
# The driver's line (python bench.py) on the box, summarised.  $1 = tag under gpurun_out/r05/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=${1:-default}; OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python bench.py ${@:2} > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "rc=$?"; tail -3 $OUT/bench_$TAG.err
python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json"))
print("step", d["ms_per_step"], "ms", d["value"], "img/s", {k:d["roofline"].get(k) for k in ("frac","frac_survey_units","frac_traffic","ms_per_launch")})
print("roi_tiles", d.get("roi_tiles"))
for k,v in d.get("extra_workloads", {}).items(): print(k, v["ms_per_step"], v.get("vs_default_step"), v.get("roi_tiles"), (v.get("roofline") or {}).get("kernels_ms"))
PY
