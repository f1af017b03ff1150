#!/bin/bash
# r06: same-box A/B of the rotated NMS mask tile on bench.py --workload rrpn_micro (profiling build: D2AMD_NMS_ROT_WAVE = r04's
# one-wave tile):  bash scripts/r06_rot_ab.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_rot_ab}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for REP in 1 2 3; do for S in wg wave:D2AMD_NMS_ROT_WAVE=1; do
  NAME=$(echo $S | cut -d: -f1); ENVS=$(echo $S | cut -s -d: -f2 | tr ',' ' ')
  env $ENVS D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so timeout 300 python bench.py --workload rrpn_micro --no-cpu-baseline > $OUT/bench_${NAME}_$REP.json 2> $OUT/bench_${NAME}_$REP.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${NAME}_$REP.json")); print("$NAME", $REP, d["ms_per_step"], d.get("roofline", {}).get("kernels_ms"))
except Exception as e: print("$NAME failed", e)
PY
done; done
