import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
plan = []
for cfg in sys.argv[1].split(";"):
    thr, ns = cfg.split(",")
    os.environ["D2AMD_FWD_THREADS"] = thr
    os.environ["D2AMD_FWD_NSPLIT"] = ns
    for name, pooler, lists in (("box", w.box_pooler, w.box_lists), ("mask", w.mask_pooler, w.mask_lists)):
        for _ in range(3):
            pooler([f.detach() for f in w.feats], lists)
        plan.append(f"{name}_thr={thr}_nsplit={ns}")
torch.cuda.synchronize()
json.dump(plan, open(os.environ.get("PLAN_OUT", "/tmp/plan.json"), "w"))
