#!/bin/bash
# round 4, second half: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, no trace domains) of the kernels
# added / changed there -> gpurun_out/$1/pmc_<op>.json (copied to profiles/r04/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${1:-pmc_r04b}; mkdir -p $OUT
N=5
cd /tmp
for OP in ${PMC_OPS:-retinanet_select pool_rot_fwd pool_rot_bwd nms_rotated iou_rotated}; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_${OP}_$CNT -o p -- python $REPO/scripts/pmc_op.py $OP nhwc $N > $OUT/pmc_${OP}_$CNT.log 2>&1; echo "pmc $OP $CNT rc=$?"
  done
  EXCL=""
  case $OP in pool_rot_bwd) EXCL="pool_rot_kernel<d2amd::bf16_t, 8, false>";; esac
  python $REPO/scripts/pmc_summary.py $OP $N $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE $EXCL at:: rocprim Cat elementwise > $OUT/pmc_$OP.json
  rm -rf $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE
  python - <<PY
import json
d = json.load(open("$OUT/pmc_$OP.json"))
kf, kw = d["kernels_fetch"], d["kernels_write"]
print("$OP", "hbm MB / launch", round(d["hbm_bytes_per_launch"] / 1e6, 1))
for k in kf:
    print("   %-70s fetch %.1f MB  write %.1f MB" % (k[:70], 2 * kf[k]["sum_KiB"] * 1024 / 1e6 / d["launches"], kw.get(k, {"sum_KiB": 0})["sum_KiB"] * 1024 / 1e6 / d["launches"]))
PY
done
