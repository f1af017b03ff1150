#!/bin/bash
# v8 GPU visit: parity of the queue-scheduled pooler backward + mask-head glue, A/B of the scheduling knobs,
# per-workgroup timelines, bench.   scripts/gpu_v8.sh <tag> [full]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD
TAG=${1:-v8}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt; nproc >> $OUT/gpu.txt
echo "== pytest"
if [ "${2:-}" = "full" ]; then SEL="tests"; else SEL="tests/test_gpu_pooler.py tests/test_gpu_mask_head.py tests/test_gpu_masks.py"; fi
timeout 1500 python -m pytest $SEL -m gpu -q --timeout=600 -p no:cacheprovider --tb=short -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log
echo "== pool bwd A/B"
{
D2AMD_POOL_NOQUEUE=1 timeout 200 python scripts/pool_bwd_ab.py noqueue
timeout 200 python scripts/pool_bwd_ab.py queue_thr4_16
D2AMD_POOL_QTHR_FINE=2 D2AMD_POOL_QTHR_COARSE=8 timeout 200 python scripts/pool_bwd_ab.py queue_thr2_8
D2AMD_POOL_QTHR_FINE=6 D2AMD_POOL_QTHR_COARSE=24 timeout 200 python scripts/pool_bwd_ab.py queue_thr6_24
D2AMD_POOL_QTHR_FINE=1000 D2AMD_POOL_QTHR_COARSE=1000 timeout 200 python scripts/pool_bwd_ab.py queue_noheavy
D2AMD_BWD_CFG=2222 timeout 200 python scripts/pool_bwd_ab.py queue_cfg2222
D2AMD_BWD_CFG=1212 timeout 200 python scripts/pool_bwd_ab.py queue_cfg1212
D2AMD_BWD_CFG=1241 timeout 200 python scripts/pool_bwd_ab.py queue_cfg1241
D2AMD_NO_SIDE_STREAM=1 timeout 200 python scripts/pool_bwd_ab.py queue_noside
} 2>&1 | grep -v Warning | tee $OUT/pool_bwd_ab.txt
echo "== timelines"
for W in box mask; do timeout 200 python scripts/pool_stamps.py $W 2>&1 | grep -v Warning; done | tee $OUT/pool_bwd_timeline.txt
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 5 --layout nhwc > $OUT/bench_nhwc.json 2> $OUT/bench_nhwc.err; echo "bench rc=$?"
cat $OUT/bench_nhwc.json; tail -3 $OUT/bench_nhwc.err
