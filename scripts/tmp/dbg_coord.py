import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from detectron2_amd import layers
from test_gpu_dcn_tc import make_case, run_oracle
for (B, C, Co, H, W) in [(1, 64, 64, 6, 7), (2, 128, 128, 18, 21), (2, 128, 128, 100, 168)]:
    for dtype in (torch.bfloat16, torch.float16):
        case = make_case(5, B, C, Co, H, W, dtype=dtype)
        x, off, msk, w, bias, go, kw = case
        exp = run_oracle(*case) if H < 50 else None
        xt = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ot, mt, wt, bt = [t.cuda().requires_grad_(True) for t in (off, msk, w, bias)]
        y = layers.modulated_deform_conv(xt, ot, mt, wt, bt, 1, 1, 1, 1, 1)
        y.backward(go.cuda().contiguous(memory_format=torch.channels_last))
        got = dict(out=y, grad_input=xt.grad, grad_offset=ot.grad, grad_mask=mt.grad, grad_weight=wt.grad, grad_bias=bt.grad)
        if exp is None:
            xt2 = x.cuda().requires_grad_(True)
            ot2, mt2, wt2, bt2 = [t.cuda().requires_grad_(True) for t in (off, msk, w, bias)]
            y2 = layers.modulated_deform_conv(xt2, ot2, mt2, wt2, bt2, 1, 1, 1, 1, 1)
            y2.backward(go.cuda())
            exp = {k: v.float().cpu().numpy() for k, v in dict(out=y2.detach(), grad_input=xt2.grad, grad_offset=ot2.grad, grad_mask=mt2.grad, grad_weight=wt2.grad, grad_bias=bt2.grad).items()}
        line = []
        for k, v in got.items():
            g = v.detach().float().cpu().numpy(); e = exp[k]
            line.append("%s %.4f" % (k, np.abs(g - e).max() / max(np.abs(e).max(), 1e-9)))
        print((B, C, Co, H, W), str(dtype)[6:], " ".join(line), flush=True)
