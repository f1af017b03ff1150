#!/bin/bash
# round 3 evidence: GPU suite, smoke, the default bench line (with extra workloads + CPU baseline), kernel stats of the
# default step, PMC traffic of the roofline op.  Everything lands in gpurun_out/$TAG (copied to profiles/r03 afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${TAG:-r3ev}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt; nproc >> $OUT/gpu.txt
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_default.json
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --disconnected > $OUT/bench_disconnected.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --layout nchw > $OUT/bench_nchw.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --no-graph > $OUT/bench_eager.json 2> /dev/null
timeout 300 python bench.py --workload retinanet_100k --no-cpu-baseline > $OUT/bench_retinanet_100k.json 2> /dev/null
timeout 300 python bench.py --workload dcn_r50 --no-cpu-baseline > $OUT/bench_dcn_r50.json 2> /dev/null
python - <<PY
import json
for n in ("default","disconnected","nchw","eager","retinanet_100k","dcn_r50"):
    try:
        d=json.load(open("$OUT/bench_%s.json"%n)); print(n, d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("frac_traffic"))
    except Exception as e: print(n,"failed",e)
PY
cd /tmp
for WL in maskrcnn_train retinanet_100k dcn_r50; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -o bench -- python $REPO/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof_$WL.log 2>&1
  cp $(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1) $OUT/${WL}_kernel_stats.csv; rm -rf $OUT/prof_$WL
done
cd $REPO
bash scripts/gpu_pmc.sh $TAG/pmc nhwc roi_align_box_bwd roi_align_mask_bwd roi_align_box_fwd > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log | cut -c1-300
