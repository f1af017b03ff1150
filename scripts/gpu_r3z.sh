#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 300 python scripts/dcn_bench.py --fwd-only --cfgs "4,1,2,1,2;4,2,2,1,2;4,2,1,1,2;4,1,4,1,2;4,2,2,1,4;4,1,2,1,4" 2>&1 | tail -3 | cut -c1-1500
