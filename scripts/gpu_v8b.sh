#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
OUT=$PWD/gpurun_out/${1:-v8b}; mkdir -p $OUT
echo "== pytest"
timeout 900 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_mask_head.py tests/test_gpu_parity.py -m gpu -q --timeout=600 -p no:cacheprovider --tb=short -x -k "${2:-pool or mask_head or roi_align}" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log
echo "== pool bwd A/B"
{
D2AMD_POOL_NOSTAGED=1 timeout 200 python scripts/pool_bwd_ab.py regs_2launch
timeout 200 python scripts/pool_bwd_ab.py staged_thr6

D2AMD_POOL_QTHR=12 timeout 200 python scripts/pool_bwd_ab.py staged_thr12

} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/pool_bwd_ab.txt
echo "== timelines"
for W in box mask; do timeout 200 python scripts/pool_stamps.py $W 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $OUT/pool_bwd_timeline.txt
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 5 --layout nhwc --no-cpu-baseline > $OUT/bench_nhwc.json 2> $OUT/bench_nhwc.err; echo "bench rc=$?"
python - $OUT/bench_nhwc.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernels_ms"])
print({k:v["ms_per_step"] for k,v in d["ops"].items()})
PY
tail -3 $OUT/bench_nhwc.err
