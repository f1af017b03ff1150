#!/bin/bash
# NMS-only GPU check: parity tests (bounded by timeout -- the reduce kernel spins on LDS flags), per-block stamps, kernel trace.
mkdir -p gpurun_out/nms; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rpn.py -q -x -k "nms or rpn" 2>&1 | tail -3
D2AMD_NMS_MASK_EXACT=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rpn.py -q -x -k "nms or rpn" 2>&1 | tail -3
timeout 120 python scripts/nms_stamps.py 2>&1 | tail -4
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nms/prof -o nms -- python $R/scripts/nms_stamps.py > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/nms/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nms' in r['Name']: print(r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'])
PY
