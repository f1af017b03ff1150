"""profiles/r06/pmc_traffic_nhwc.json from the per-op summaries of scripts/pmc_r06.sh (HBM bytes per launch; FETCH_SIZE x2
on gfx950, WRITE_SIZE as reported: scripts/pmc_summary.py).  bench.py's `roofline.traffic` reads it.
    python scripts/pmc_r06_summary.py <dir with pmc_*.json> > profiles/r06/pmc_traffic_nhwc.json"""
import json, os, sys

d = sys.argv[1]
load = lambda op: json.load(open(os.path.join(d, "pmc_%s.json" % op)))


def kernel_bytes(j, sub):
    n = j["launches"]
    f = sum(v["sum_KiB"] for k, v in j["kernels_fetch"].items() if sub in k)
    w = sum(v["sum_KiB"] for k, v in j["kernels_write"].items() if sub in k)
    return int((2.0 * f + w) * 1024 / n)


ops = {}
cs = load("connected_step")
ops["roi_align_pair_bwd"] = {
    "hbm_bytes_per_launch": kernel_bytes(cs, "roi_records_kernel") + kernel_bytes(cs, "tile_lists_kernel") + kernel_bytes(cs, "pool_bwd_kcat_kernel"),
    "kernel_hbm_bytes_per_launch": kernel_bytes(cs, "pool_bwd_kcat_kernel"),
    "note": "the CONNECTED step's own sampled ROI lists (scripts/pmc_op.py connected_step): records + tile lists + the paired K-concatenated tile gather (pool_bwd_kcat_kernel)"}
ops["roi_align_pair_fwd"] = {"hbm_bytes_per_launch": kernel_bytes(cs, "pool_fwd_nhwc_kernel"),
                             "note": "pool_fwd_nhwc_kernel of the connected step (both poolers, one launch)"}
ops["connected_step_all_kernels"] = {"hbm_bytes_per_launch": cs["hbm_bytes_per_launch"]}
for op in ("paste_masks", "iou_rotated", "retinanet_select"):  # (r04's figures re-measured at HEAD)
    try:
        ops[op] = {"hbm_bytes_per_launch": load(op)["hbm_bytes_per_launch"]}
    except Exception as e:
        ops[op] = {"error": str(e)}
stages = {"res3": (load("dcn_bwd_res3"), 4), "res4": (load("dcn_bwd_res4"), 6), "res5": (load("dcn_bwd_res5"), 3)}
names = {"dcn_fwd_col": "dcn_col_kernel", "dcn_bwd_coord": "dcn_coord_grad_kernel", "dcn_bwd_gather": "dcn_gather_dx_kernel",
         "dcn_bwd_weight": "dcn_bww_gemm_kernel", "dcn_bwd_weight_reduce": "bww_gemm_reduce_kernel", "dcn_bin_samples": "dcn_bin_samples_kernel",
         "dcn_sort_lists": "dcn_sort_lists_kernel"}
for op, sub in names.items():
    per = {t: kernel_bytes(j, sub) for t, (j, _n) in stages.items()}
    ops[op] = {"hbm_bytes_per_launch": int(sum(per[t] * n for t, (_j, n) in stages.items()) / 13), "per_stage": per}
# the two GEMMs share the kernel template: told apart by their write size (forward writes P x Co, backward-data P x 9C)
for op, big in (("dcn_fwd_gemm", False), ("dcn_bwd_dcol_gemm", True)):
    per = {}
    for t, (j, _n) in stages.items():
        ks = [k for k in j["kernels_fetch"] if "gemm_nt_kernel" in k]
        ks.sort(key=lambda k: j["kernels_write"].get(k, {"sum_KiB": 0})["sum_KiB"])
        k = ks[-1] if big else ks[0]
        per[t] = int((2.0 * j["kernels_fetch"][k]["sum_KiB"] + j["kernels_write"].get(k, {"sum_KiB": 0})["sum_KiB"]) * 1024 / j["launches"])
    ops[op] = {"hbm_bytes_per_launch": int(sum(per[t] * n for t, (_j, n) in stages.items()) / 13), "per_stage": per}
print(json.dumps({"source": "scripts/pmc_r06.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, 5 launches per op; "
                            "FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE as reported; DCN ops: mean over the 13 R50 "
                            "blocks (4 res3 + 6 res4 + 3 res5)", "layout": "nhwc", "dtype": "bf16", "ops": ops}, indent=1))
