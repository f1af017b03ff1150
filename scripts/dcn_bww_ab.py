"""Kernel time of the DCN weight-gradient kernel per R50 stage (library events) under the current environment
(D2AMD_DCN_BWW_COOP, D2AMD_DCN_BWW_PCH).  python scripts/dcn_bww_ab.py [tag]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd import _C
from detectron2_amd.layers import ModulatedDeformConv
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
out = []
for name, (C, H, W) in (("res3", (128, 100, 168)), ("res4", (256, 50, 84)), ("res5", (512, 25, 42))):
    mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def step():
        y = mod(x, off, msk)
        y.backward(torch.ones_like(y))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _C.lib().d2amd_timing_select(b"dcn_bwd_weight")
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    tot, n = ctypes.c_double(0), ctypes.c_int(0)
    r = _C.lib().d2amd_timing_read(b"dcn_bwd_weight", ctypes.byref(tot), ctypes.byref(n))
    _C.lib().d2amd_timing_select(None)
    out.append(f"{name} {tot.value / max(n.value, 1) * 1e3:.1f} us")
print(f"[{tag}] " + " | ".join(out), flush=True)
