#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
D2AMD_POOL_FWD_MFMA=0 timeout 120 python scripts/pool_fwd_mfma_stamps.py box 2>&1 | grep -v amdgpu.ids
D2AMD_POOL_FWD_MFMA=1 timeout 120 python scripts/pool_fwd_mfma_stamps.py box 2>&1 | grep -v amdgpu.ids
D2AMD_POOL_FWD_MFMA=1 timeout 120 python scripts/pool_fwd_mfma_stamps.py mask 2>&1 | grep -v amdgpu.ids
