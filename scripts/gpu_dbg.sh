#!/bin/bash
# per-phase cycle stamps of single tile workgroups (needs a -DD2AMD_PROFILE build of the library)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
OUT=$PWD/gpurun_out/${1:-dbg}; mkdir -p $OUT
timeout 300 python scripts/exp_dbg.py 0 3 8 64 400 800 1200 1600 2>&1 | grep "d2amd dbg" | tee -a $OUT/stamps.txt
