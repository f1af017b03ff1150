#!/bin/bash
# round 5: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, no trace domains) at HEAD of the headline's roofline
# kernels (paired pooler backward / forward) and of the rebuilt DCN path (column, GEMMs, coordinate gradient, list sort,
# gather) -> gpurun_out/$1/pmc_<op>.json (copied to profiles/r05/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/${1:-pmc_r05}; mkdir -p $OUT
N=5
cd /tmp
for OP in ${PMC_OPS:-roi_align_chain_bwd roi_align_pair_fwd dcn_bwd_res3 dcn_bwd_res4 dcn_bwd_res5}; do
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
