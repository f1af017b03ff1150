"""Ablation of the DCN MFMA kernels (profiling switches D2AMD_DCN_ABLATE / _BWD, D2AMD_DCN_PATCH_R,
D2AMD_DCN_CFG).  Run under `rocprofv3 --kernel-trace`; prints the plan (JSON list of labels, REP
dispatches each, in launch order) that scripts/dcn_ablate_parse.py matches with the trace."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.layers import ModulatedDeformConv

REP = 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
fwd_plan, bwd_plan = [], []
shapes = {"res3": (128, 100, 168), "res4": (256, 50, 84), "res5": (512, 25, 42)}
for tag in sys.argv[1:] or ["res3"]:
    C, H, W = shapes[tag]
    mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
    x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16, requires_grad=True)
    off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
    msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16).requires_grad_(True)
    for ab in (0, 1, 2, 3, 4, 8, 12, 7, 15):
        os.environ["D2AMD_DCN_ABLATE"] = str(ab)
        for _ in range(REP):
            mod(x.detach(), off.detach(), msk.detach())
        fwd_plan.append(f"fwd_{tag}_ab{ab}")
    os.environ["D2AMD_DCN_ABLATE"] = "0"
    y = mod(x, off, msk)
    fwd_plan.append(f"fwd_{tag}_single")
    g = torch.randn_like(y)
    for R in ("-1",):
        os.environ["D2AMD_DCN_PATCH_R"] = R
        for ab in (0, 1, 2, 4, 16, 6, 7):
            os.environ["D2AMD_DCN_ABLATE_BWD"] = str(ab)
            for _ in range(REP):
                torch.autograd.grad([y], [x, off, msk], [g], retain_graph=True)
            bwd_plan.append(f"bwd_{tag}_R{R}_ab{ab}")
    os.environ["D2AMD_DCN_ABLATE_BWD"] = "0"
    os.environ.pop("D2AMD_DCN_PATCH_R")
    torch.cuda.synchronize()
json.dump({"rep": REP, "fwd": fwd_plan, "bwd": bwd_plan}, open(os.environ.get("PLAN_OUT", "/tmp/dcn_plan.json"), "w"))
