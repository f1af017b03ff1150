#!/bin/bash
# r06: stamps (pair) + cycle stamps of single workgroup-tiles of the K-concatenated tile gather (profiling build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_dbg}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so
for K in ${KCATS:-1 3}; do
  echo "=== kcat $K pair"; D2AMD_POOL_KCAT=$K timeout 300 python scripts/pool_stamps.py pair 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/stamps_kcat${K}_pair.txt
  for B in ${BLOCKS:-24 320 1200 2400}; do
    echo "=== kcat $K dbg block $B"; D2AMD_DBG_BLOCK=$B D2AMD_POOL_KCAT=$K timeout 300 python scripts/pool_stamps.py pair 2>&1 | grep "d2amd dbg" | tail -2 | tee -a $OUT/dbg_kcat${K}.txt
  done
done
