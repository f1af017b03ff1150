#!/bin/bash
# round 4: ablation of the wave-specialised DCN backward-data kernel (D2AMD_DCN_ABLATE_BWD bits: 1 no MFMA section,
# 2 no phase A, 8 no column store, 32 no corner gathers) -- kernel time per block from the library's events
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r4h}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for AB in 0 1 42 43; do
  D2AMD_DCN_ABLATE_BWD=$AB timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline --steps 10 > $OUT/b_$AB.json 2> /dev/null
done
python - <<PY
import json
for n in (0,1,42,43):
    try:
        d=json.load(open("$OUT/b_%d.json"%n)); print("ablate %2d"%n, d["ms_per_step"], d["roofline"]["kernels_ms"], {k:v["ms_per_step"] for k,v in d["ops"].items() if k.startswith("bwd")})
    except Exception as e: print(n,"failed",e)
PY
