#!/bin/bash
# DCN evidence for profiles/: kernel stats of scripts/dcn_bench.py, forward config sweep, forward
# per-workgroup timeline, backward ablation, backward channel-split sweep.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/dcnprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o dcn -- python $REPO/scripts/dcn_bench.py > $OUT/dcn_bench_under_rocprof.json 2> $OUT/err.log
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cut -c1-260 $f | head -24 > $OUT/dcn_kernel_stats.csv
python $REPO/scripts/trace_seq.py $(find $OUT/prof -name "*kernel_trace.csv") dcn_fwd_tc dcn_bwd_data_tc dcn_bwd_weight > $OUT/dcn_kernel_durations_by_shape.txt
rm -rf $OUT/prof
cd $REPO
python scripts/dcn_bench.py > $OUT/dcn_bench.json 2>> $OUT/err.log
bash scripts/gpu_sweep.sh fwd D2AMD_DCN_CFG "4,1,4,1,4;4,1,2,1,4;4,1,1,1,4;4,2,2,1,4;4,2,1,1,4;4,1,4,1,2;4,1,2,1,2;4,1,1,1,2;4,1,2,2,2;4,1,2,4,2;4,1,1,4,2;4,1,1,1,2,1;4,1,1,3,2,1;4,1,1,6,2,1" 2>&1 | grep "fwd_" > $OUT/dcn_fwd_sweep.txt
bash scripts/gpu_sweep.sh bwd D2AMD_DCN_CSPLIT "1;2;4;8" 2>&1 | grep "bwd_" > $OUT/dcn_bwd_csplit_sweep.txt
bash scripts/gpu_sweep.sh bwd D2AMD_DCN_ABLATE_BWD "0;1;2;4;16;6;7" res3 2>&1 | grep "bwd_" > $OUT/dcn_bwd_ablation_res3.txt
for t in res3 res4 res5; do python scripts/dcn_stamps.py $t 2>/dev/null; done > $OUT/dcn_fwd_timeline.txt
D2AMD_DCN_CFG=4,1,4,1,4 python scripts/dcn_stamps.py res3 2>/dev/null > $OUT/dcn_fwd_timeline_res3_cfg4141_64ch.txt
python scripts/matcher_bench.py 2>/dev/null | tail -1 > $OUT/matcher_bench.json
python scripts/nms_overlap.py 2>/dev/null | grep ms > $OUT/nms_images_overlap.txt
rm -rf $REPO/gpurun_out/sweep
ls -la $OUT
