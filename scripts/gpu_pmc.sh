#!/bin/bash
# HBM traffic of single ops from PMC counters; separate passes per counter (no trace domains).
#   scripts/gpu_pmc.sh <tag> <layout> <op> [<op> ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; TAG=$1; LAYOUT=$2; shift 2
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
N=5
cd /tmp
for OP in "$@"; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_${OP}_$CNT -o p -- python $REPO/scripts/pmc_op.py $OP $LAYOUT $N > $OUT/pmc_${OP}_$CNT.log 2>&1; echo "pmc $OP $CNT rc=$?"
  done
  EXCL=""
  case $OP in *_bwd) EXCL="pool_fwd boxes_to_rois";; dcn_bwd*) EXCL="dcn_fwd tc_pack_weight_kernel";; esac
  python $REPO/scripts/pmc_summary.py $OP $N $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE $EXCL at:: rocprim Cat elementwise > $OUT/pmc_$OP.json
  cat $OUT/pmc_$OP.json | cut -c1-600
  rm -rf $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE
done
