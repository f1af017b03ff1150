"""DCNv2 device time at the three R50 stage shapes of SURVEY 8(a) a5/a6 (bf16, batch 2), optionally for a
list of forward tile configurations (D2AMD_DCN_CFG values).  Prints one JSON line.

    python scripts/dcn_bench.py [--cfgs "4,1,4,1;4,1,2,1"] [--v1] [--fwd-only]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.layers import ModulatedDeformConv

MFMA_BF16 = 2500.0


def timeit(fn, rep=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / rep


def main():
    dev = torch.device("cuda", 0)
    cfgs = [None]
    if "--cfgs" in sys.argv:
        cfgs = sys.argv[sys.argv.index("--cfgs") + 1].split(";")
    if "--v1" in sys.argv:
        os.environ["D2AMD_DCN_V1"] = "1"
    fwd_only = "--fwd-only" in sys.argv
    out = {}
    torch.manual_seed(0)
    for tag, (C, H, W) in (("res3", (128, 100, 168)), ("res4", (256, 50, 84)), ("res5", (512, 25, 42))):
        mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
        x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16, requires_grad=True)
        off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
        msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16).requires_grad_(True)
        flops = 2.0 * C * C * 9 * 2 * H * W
        for cfg in cfgs:
            if cfg:
                os.environ["D2AMD_DCN_CFG"] = cfg
            ms = timeit(lambda: mod(x.detach(), off.detach(), msk.detach()))
            out[f"fwd_{tag}" + (f"[{cfg}]" if cfg else "")] = {
                "ms": round(ms, 4), "TFLOPs": round(flops / 1e9 / ms, 1),
                "frac_mfma_bf16": round(flops / 1e9 / ms / MFMA_BF16, 4)}
        os.environ.pop("D2AMD_DCN_CFG", None)
        if not fwd_only:
            y = mod(x, off, msk)
            g = torch.randn_like(y)
            ms = timeit(lambda: torch.autograd.grad([y], [x, off, msk, mod.weight], [g], retain_graph=True), rep=10)
            out[f"bwd_{tag}"] = {"ms": round(ms, 4), "TFLOPs": round(2 * flops / 1e9 / ms, 1),
                                 "frac_mfma_bf16": round(2 * flops / 1e9 / ms / MFMA_BF16, 4)}
            ms = timeit(lambda: torch.autograd.grad([y], [x, off, msk], [g], retain_graph=True), rep=10)
            out[f"bwd_data_only_{tag}"] = {"ms": round(ms, 4)}
            del y
    print(json.dumps(out))


if __name__ == "__main__":
    main()
