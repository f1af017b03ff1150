#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/v5prof; rm -rf $OUT; mkdir -p $OUT
for w in box mask; do python scripts/pool_stamps.py $w 2>/dev/null; done > $OUT/pool_bwd_timeline.txt
for w in box mask; do python scripts/pool_fwd_stamps.py $w 2>/dev/null; done > $OUT/pool_fwd_timeline.txt
export PLAN_OUT=/tmp/plan.json
cd /tmp; rm -rf /tmp/p
rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o k -- python $REPO/scripts/pool_fwd_sweep.py "256,4;256,16;512,1;512,2;512,4;1024,1;1024,4" > /dev/null 2>&1
python - > $OUT/pool_fwd_sweep.txt <<PY
import csv,glob,json
plan=json.load(open("/tmp/plan.json"))
rows=list(csv.DictReader(open(glob.glob("/tmp/p/**/*kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "pool_fwd_nhwc" in r["Kernel_Name"]]
for i,l in enumerate(plan): print(l, [round(v,1) for v in d[3*i:3*i+3]], "us")
PY
cd $REPO
bash scripts/gpu_sweep.sh bww D2AMD_DCN_BWW_PCH "0;8;16;32;64;128" 2>&1 | grep "bww_" > $OUT/dcn_bww_sweep.txt
bash scripts/gpu_sweep.sh fwd D2AMD_DCN_NONE "0" 2>&1 | grep "fwd_" > $OUT/dcn_default_kernel_times.txt
bash scripts/gpu_sweep.sh bwd D2AMD_DCN_NONE "0" 2>&1 | grep "bwd_" >> $OUT/dcn_default_kernel_times.txt
bash scripts/gpu_sweep.sh bww D2AMD_DCN_NONE "0" 2>&1 | grep "bww_" >> $OUT/dcn_default_kernel_times.txt
python scripts/rpn_bench.py 2>/dev/null | tail -1 > $OUT/rpn_bench.json
python scripts/matcher_bench.py 2>/dev/null | tail -1 > $OUT/matcher_bench.json
python scripts/host_profile.py 2>/dev/null | head -3 > $OUT/host_wall_per_step.txt
rm -rf $REPO/gpurun_out/sweep
ls $OUT
