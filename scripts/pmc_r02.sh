#!/bin/bash
# round 2: HBM traffic (PMC, separate FETCH_SIZE / WRITE_SIZE passes, no trace domains) of the chained pooler backward
# and the forward poolers -> gpurun_out/$1/pmc_traffic_nhwc.json in the format bench.py reads (per kernel of the op)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD; OUT=$REPO/gpurun_out/${1:-pmc_r02}; mkdir -p $OUT
N=5
cd /tmp
for OP in roi_align_chain_bwd roi_align_box_fwd roi_align_mask_fwd mask_targets; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_${OP}_$CNT -o p -- python $REPO/scripts/pmc_op.py $OP nhwc $N > $OUT/pmc_${OP}_$CNT.log 2>&1; echo "pmc $OP $CNT rc=$?"
  done
  python $REPO/scripts/pmc_summary.py $OP $N $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE at:: rocprim Cat elementwise > $OUT/pmc_$OP.json
  rm -rf $OUT/pmc_${OP}_FETCH_SIZE $OUT/pmc_${OP}_WRITE_SIZE
done
python - <<PY
import json
N = $N
out = {"source": "scripts/pmc_r02.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d launches per op; KiB; FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE as reported" % N,
       "layout": "nhwc", "dtype": "bf16", "ops": {}}
def per_kernel(d, sub):
    f = sum(v["sum_KiB"] for k, v in d["kernels_fetch"].items() if sub in k) / N
    w = sum(v["sum_KiB"] for k, v in d["kernels_write"].items() if sub in k) / N
    return {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
d = json.load(open("$OUT/pmc_roi_align_chain_bwd.json"))
# the roofline kernel alone: the 7x7 tile gather (head of the chain)
out["ops"]["roi_align_box_bwd"] = dict(per_kernel(d, "pool_bwd_mfma_kernel<d2amd::bf16_t, 8>"), note="pool_bwd_mfma_kernel<bf16_t, 8> alone, inside the chained backward (the head of the chain: plain write of every non-empty tile; empty tiles are zero-filled by tile_lists_kernel, not counted here)")
out["ops"]["roi_align_mask_bwd"] = dict(per_kernel(d, "pool_bwd_mfma_kernel<d2amd::bf16_t, 16>"), note="pool_bwd_mfma_kernel<bf16_t, 16> alone, accumulate mode (adds to the box pooler's gradient; tiles no ROI touches are neither read nor written)")
out["ops"]["backward_poolers_all_kernels"] = {k: d[k] for k in ("FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch", "hbm_bytes_per_launch")}
for op in ("roi_align_box_fwd", "roi_align_mask_fwd", "mask_targets"):
    e = json.load(open("$OUT/pmc_%s.json" % op))
    out["ops"][op] = {k: e[k] for k in ("FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch", "hbm_bytes_per_launch")}
json.dump(out, open("$OUT/pmc_traffic_nhwc.json", "w"), indent=1)
print(json.dumps(out["ops"], indent=1)[:1800])
PY
