"""Generic env-switch sweep for the DCN kernels under rocprofv3 --kernel-trace.
    python scripts/dcn_sweep.py <fwd|bwd> <ENV_NAME> <v1,v2,...> [shapes...]
Writes the plan for scripts/dcn_ablate_parse.py."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.layers import ModulatedDeformConv

REP = 3
mode, name, vals = sys.argv[1], sys.argv[2], sys.argv[3].split(";")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
plan = {"rep": REP, "fwd": [], "bwd": [], "bww": []}
shapes = {"res3": (128, 100, 168), "res4": (256, 50, 84), "res5": (512, 25, 42)}
for tag in sys.argv[4:] or ["res3", "res4", "res5"]:
    C, H, W = shapes[tag]
    mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
    x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16, requires_grad=True)
    off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
    msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16).requires_grad_(True)
    if mode in ("bwd", "bww"):
        y = mod(x, off, msk)
        plan["fwd"].append(f"fwd_{tag}_single")
        g = torch.randn_like(y)
    for v in vals:
        os.environ[name] = v
        for _ in range(REP):
            if mode == "fwd":
                mod(x.detach(), off.detach(), msk.detach())
            elif mode == "bwd":
                torch.autograd.grad([y], [x, off, msk], [g], retain_graph=True)
            else:  # bww: weight gradient only
                torch.autograd.grad([y], [mod.weight], [g], retain_graph=True)
        plan[mode].append(f"{mode}_{tag}_{name}={v}")
    os.environ.pop(name)
    torch.cuda.synchronize()
json.dump(plan, open(os.environ.get("PLAN_OUT", "/tmp/dcn_plan.json"), "w"))
