#!/bin/bash
# r06: per-workgroup stamps of the K-concatenated tile gather (profiling build): box / mask / pair per KCAT setting.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_stamps}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export D2AMD_LIB_PATH=$REPO/detectron2_amd/lib/libd2amd_prof.so
for K in ${KCATS:-1 3}; do for W in box mask pair; do
  echo "=== kcat $K $W"; D2AMD_POOL_KCAT=$K timeout 300 python scripts/pool_stamps.py $W 2>&1 | grep -v Warning | tee $OUT/stamps_kcat${K}_$W.txt
  echo "=== kcat $K $W EMPTY"; D2AMD_ABLATE=64 D2AMD_POOL_KCAT=$K timeout 300 python scripts/pool_stamps.py $W 2>&1 | grep -v Warning | tee $OUT/stamps_kcat${K}_${W}_empty.txt
done; done
