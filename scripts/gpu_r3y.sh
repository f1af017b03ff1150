#!/bin/bash
# SQ / cache counters of one op of the step (which pipe does the kernel wait for?)   usage: OP=roi_align_box_fwd KEY=pool_fwd bash scripts/gpu_r3y.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${TAG:-r3y}; mkdir -p $OUT
OP=${OP:-roi_align_box_fwd}; KEY=${KEY:-pool_fwd}
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_REQ_sum"
P5="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
n=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/sq$n -o p -- python $REPO/scripts/pmc_op.py $OP nhwc 3 > $OUT/sq$n.log 2>&1; echo "pass $n rc=$?"
  f=$(find $OUT/sq$n -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { tail -3 $OUT/sq$n.log; continue; }
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:60]
    if "$KEY" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in acc:
    print(k)
    for c,v in acc[k].items(): print("   %-34s %16.0f per launch (%d launches)"%(c, v/cnt[(k,c)], cnt[(k,c)]))
PY
  rm -rf $OUT/sq$n
done
