#!/bin/bash
# Same-box A/B of two builds of the library (D2AMD_LIB_PATH): the default bench line (+ extra workloads) per build.
#   gpurun -- 'bash scripts/lib_ab.sh TAG name1:/path/to/lib1.so name2:/path/to/lib2.so'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-lib_ab}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; shift
for REP in 1 2; do for S in "$@"; do
  NAME=${S%%:*}; LIB=${S#*:}
  D2AMD_LIB_PATH=$REPO/$LIB timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_${NAME}_$REP.json 2> $OUT/bench_${NAME}_$REP.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_${NAME}_$REP.json"))
ew = d.get("extra_workloads", {})
print("$NAME", $REP, d["ms_per_step"], d["roofline"]["kernels_ms"], {k: v.get("ms_per_step") for k, v in ew.items()})
PY
done; done
