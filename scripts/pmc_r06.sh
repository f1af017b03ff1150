#!/bin/bash
# round 6: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, no trace domains) at HEAD of every op bench.py quotes
# `roofline.traffic` for: the connected step (paired pooler backward = K-concatenated tile gather, paired forward), the paste,
# the rotated IoU, the dense-detector selection, and the DCN column path per stage -> gpurun_out/$1/pmc_<op>.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${1:-pmc_r06}; mkdir -p $OUT
N=5
cd /tmp
for OP in ${PMC_OPS:-connected_step paste_masks iou_rotated retinanet_select dcn_bwd_res3 dcn_bwd_res4 dcn_bwd_res5}; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_${OP}_$CNT -o p -- python $REPO/scripts/pmc_op.py $OP nhwc $N > $OUT/pmc_${OP}_$CNT.log 2>&1; echo "pmc $OP $CNT rc=$?"
  done
  python $REPO/scripts/pmc_summary.py $OP $N $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE at:: rocprim Cat elementwise > $OUT/pmc_$OP.json
  rm -rf $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE
  python - <<PY
import json
d = json.load(open("$OUT/pmc_$OP.json"))
kf, kw = d["kernels_fetch"], d["kernels_write"]
print("$OP", "hbm MB / launch", round(d["hbm_bytes_per_launch"] / 1e6, 1))
for k in kf:
    print("   %-80s fetch %.1f MB  write %.1f MB" % (k[:80], 2 * kf[k]["sum_KiB"] * 1024 / 1e6 / d["launches"], kw.get(k, {"sum_KiB": 0})["sum_KiB"] * 1024 / 1e6 / d["launches"]))
PY
done
python $REPO/scripts/pmc_r06_summary.py $OUT > $OUT/pmc_traffic_nhwc.json && echo "wrote $OUT/pmc_traffic_nhwc.json"
