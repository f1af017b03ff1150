#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
OUT=$PWD/gpurun_out/${1:-ab}; mkdir -p $OUT
{
timeout 200 python scripts/pool_bwd_ab.py "${1:-ab}"
timeout 200 python scripts/pool_bwd_ab.py "${1:-ab}_again"
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/pool_bwd_ab.txt
for W in box mask; do timeout 200 python scripts/pool_stamps.py $W 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $OUT/pool_bwd_timeline.txt
