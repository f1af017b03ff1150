import csv, json, sys
plan = json.load(open(sys.argv[1]))
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def durs(nm):
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if nm in r['Kernel_Name']]
f = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "dcn_fwd_tc_kernel" in r["Kernel_Name"] or "dcn_fwd_wave_kernel" in r["Kernel_Name"]]
b = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "dcn_bwd_data_tc_kernel" in r["Kernel_Name"] or "dcn_bwd_data_patch_kernel" in r["Kernel_Name"]]
rep = plan["rep"]
i = 0
for lab in plan["fwd"]:
    n = 1 if lab.endswith("single") else rep
    print(lab, [round(v, 1) for v in f[i:i + n]]); i += n
i = 0
for lab in plan["bwd"]:
    print(lab, [round(v, 1) for v in b[i:i + rep]]); i += rep
i = 0
w = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "dcn_bwd_weight" in r["Kernel_Name"]]
for lab in plan.get("bww", []):
    print(lab, [round(v, 1) for v in w[i:i + rep]]); i += rep
