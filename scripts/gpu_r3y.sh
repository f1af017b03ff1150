#!/bin/bash
# SQ counters of the box pooler backward (is the tile gather VALU-issue bound?), RCCL capture-mode check, extra_workloads check
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${TAG:-r3y}; mkdir -p $OUT
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_dist.py -q -p no:cacheprovider 2>&1 | tail -1; done
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json")); print("default", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["extra_workloads"].items()})
PY
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P3="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
n=0
for P in "$P1" "$P2" "$P3"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/sq$n -o p -- python $REPO/scripts/pmc_op.py roi_align_box_bwd nhwc 3 > $OUT/sq$n.log 2>&1; echo "pass $n rc=$?"
  f=$(find $OUT/sq$n -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:60]
    if "pool_bwd" not in k and "tile_lists" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in acc:
    print(k)
    for c,v in acc[k].items(): print("   %-34s %16.0f per launch (%d launches)"%(c, v/cnt[(k,c)], cnt[(k,c)]))
PY
  rm -rf $OUT/sq$n
done
