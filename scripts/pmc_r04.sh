#!/bin/bash
# round 4: PMC evidence (separate passes per counter set, no trace domains -- MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   * HBM traffic (FETCH_SIZE, WRITE_SIZE) of the pooler ops (bench.py's roofline.traffic), the DCN kernels at the three
#     R50 stage shapes, paste_masks, the rotated IoU
#   * SQ counters of the DCN kernels: VALU / MFMA busy, wait fractions, LDS bank conflicts
# -> gpurun_out/$1/{pmc_traffic_nhwc.json, dcn_sq_counters.json}  (copied to profiles/r04/ afterwards)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/${1:-pmc_r04}; mkdir -p $OUT
N=5
cd /tmp
OPS="${PMC_OPS:-roi_align_chain_bwd roi_align_box_fwd roi_align_mask_fwd paste_masks iou_rotated dcn_fwd_res3 dcn_bwd_res3 dcn_fwd_res4 dcn_bwd_res4 dcn_fwd_res5 dcn_bwd_res5}"
for OP in $OPS; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_${OP}_$CNT -o p -- python $REPO/scripts/pmc_op.py $OP nhwc $N > $OUT/pmc_${OP}_$CNT.log 2>&1; echo "pmc $OP $CNT rc=$?"
  done
  EXCL=""
  case $OP in dcn_bwd*) EXCL="dcn_fwd_tc tc_pack_weight_kernel tc_reduce_partial Random randn normal distribution";; esac
  python $REPO/scripts/pmc_summary.py $OP $N $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE $EXCL at:: rocprim Cat elementwise > $OUT/pmc_$OP.json
  rm -rf $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE
done
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
for OP in dcn_fwd_res3 dcn_bwd_res3 dcn_bwd_res4; do
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/sq_$OP -o p -- python $REPO/scripts/pmc_op.py $OP nhwc $N > $OUT/sq_$OP.log 2>&1; echo "sq $OP rc=$?"
done
python $REPO/scripts/pmc_r04_summary.py $OUT $N
