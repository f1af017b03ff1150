"""Summary of scripts/pmc_r04.sh: per-op / per-kernel HBM bytes (pmc_traffic_nhwc.json, the file bench.py reads for
roofline.traffic) and the SQ counters of the DCN kernels (dcn_sq_counters.json).   pmc_r04_summary.py <dir> <N>"""
import csv, glob, json, os
from collections import defaultdict
import sys
N = int(sys.argv[2])
OUT = sys.argv[1]
def per_kernel(d, sub):
    f = sum(v["sum_KiB"] for k, v in d["kernels_fetch"].items() if sub in k) / N
    w = sum(v["sum_KiB"] for k, v in d["kernels_write"].items() if sub in k) / N
    return {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
out = {"source": "scripts/pmc_r04.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d launches per op; KiB; "
                 "FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE as reported" % N,
       "layout": "nhwc", "dtype": "bf16", "ops": {}}
d = json.load(open(OUT + "/pmc_roi_align_chain_bwd.json"))
out["ops"]["roi_align_box_bwd"] = dict(per_kernel(d, "pool_bwd_mfma_kernel<d2amd::bf16_t, 8"), note="pool_bwd_mfma_kernel<bf16_t, 8> alone, inside the chained backward (head of the chain)")
out["ops"]["roi_align_mask_bwd"] = dict(per_kernel(d, "pool_bwd_mfma_kernel<d2amd::bf16_t, 16"), note="pool_bwd_mfma_kernel<bf16_t, 16> alone, accumulate mode")
out["ops"]["backward_poolers_all_kernels"] = {k: d[k] for k in ("FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch", "hbm_bytes_per_launch")}
for op in ("roi_align_box_fwd", "roi_align_mask_fwd", "paste_masks", "iou_rotated"):
    e = json.load(open(OUT + "/pmc_%s.json" % op))
    out["ops"][op] = {k: e[k] for k in ("FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch", "hbm_bytes_per_launch")}
# DCN kernels: per stage shape and the mean over the 13 blocks of R50 (4 x res3, 6 x res4, 3 x res5)
KER = {"dcn_fwd": ("dcn_fwd_tc_kernel", "fwd"), "dcn_bwd_data": ("dcn_bwd_data_", "bwd"), "dcn_bwd_gather": ("dcn_gather_dx_kernel", "bwd"),
       "dcn_bwd_weight": ("dcn_bww_gemm_kernel", "bwd")}
wts = {"res3": 4, "res4": 6, "res5": 3}
for key, (sub, which) in KER.items():
    per = {}
    for st in wts:
        e = json.load(open(OUT + "/pmc_dcn_%s_%s.json" % (which, st)))
        per[st] = per_kernel(e, sub)
    mean = sum(per[st]["hbm_bytes_per_launch"] * wts[st] for st in wts) / 13.0
    out["ops"][key] = {"hbm_bytes_per_launch": int(mean), "per_stage": per, "note": "mean over the 13 R50 blocks (4 res3, 6 res4, 3 res5), kernel `%s*` alone" % sub}
json.dump(out, open(OUT + "/pmc_traffic_nhwc.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out["ops"].items()}))
# SQ counters per kernel
sq = {}
for op in ("dcn_fwd_res3", "dcn_bwd_res3", "dcn_bwd_res4"):
    per = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(set)
    for f in glob.glob(os.path.join(OUT, "sq_" + op, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:70]
            if "d2amd::dcn" not in k and "bww" not in k:
                continue
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            nd[k].add(row.get("Dispatch_Id"))
    sq[op] = {}
    for k, c in per.items():
        n = max(len(nd[k]), 1)
        e = {kk: vv / n for kk, vv in c.items()}
        wc = e.get("SQ_WAVE_CYCLES", 0) or 1
        e["dispatches"] = n
        e["frac_wave_cycles_issuing (ACTIVE_INST_ANY / WAVE_CYCLES)"] = round(e.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4)
        e["frac_wave_cycles_valu (ACTIVE_INST_VALU / WAVE_CYCLES)"] = round(e.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4)
        e["frac_wave_cycles_parked (WAIT_ANY / WAVE_CYCLES)"] = round(e.get("SQ_WAIT_ANY", 0) / wc, 4)
        e["frac_wave_cycles_issue_stall (WAIT_INST_ANY / WAVE_CYCLES)"] = round(e.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
        bc = e.get("SQ_BUSY_CYCLES", 0) or 1
        e["mfma_busy_cycles_per_sq_busy_cycle (VALU_MFMA_BUSY_CYCLES / BUSY_CYCLES)"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / bc, 4)
        sq[op][k] = e
    import shutil
    shutil.rmtree(os.path.join(OUT, "sq_" + op), ignore_errors=True)
json.dump({"source": "scripts/pmc_r04.sh: one rocprofv3 --pmc pass with 8 SQ counters per op, %d launches; values are means per dispatch; "
                     "WAVE_CYCLES / WAIT_* / ACTIVE_INST_* count quad-cycles, VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)" % N,
           "ops": sq}, open(OUT + "/dcn_sq_counters.json", "w"), indent=1)
for op, ks in sq.items():
    for k, e in ks.items():
        print(op, k[:50], {kk: vv for kk, vv in e.items() if kk.startswith("frac") or kk.startswith("mfma")})