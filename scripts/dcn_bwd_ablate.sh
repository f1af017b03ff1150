#!/bin/bash
# Ablation of the wave-specialised DCN backward-data kernel: D2AMD_DCN_ABLATE_BWD bits 1 = no operand loads / MFMAs,
# 2 = no phase A, 8 = no column store, 32 = no corner gathers (42 = matrix side only, 43 = skeleton).  Kernel time per
# block from the library's launch-stream events.   gpurun -- 'bash scripts/dcn_bwd_ablate.sh TAG [bits ...]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-dcn_ablate}; shift; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
BITS="${@:-0 1 2 8 32 42 43}"
for AB in $BITS; do
  D2AMD_DCN_ABLATE_BWD=$AB timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline --steps 10 > $OUT/b_$AB.json 2> /dev/null
done
python - $BITS <<PY
import json, sys
for n in sys.argv[1:]:
    try:
        d=json.load(open("$OUT/b_%s.json"%n)); print("ablate %3s"%n, d["roofline"]["kernels_ms"])
    except Exception as e: print(n,"failed",e)
PY
