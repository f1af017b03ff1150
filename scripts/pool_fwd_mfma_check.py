"""Forward pooler: dump outputs of a set of cases (bench lists + random boxes incl. tiny / huge / outside) to a file;
run once with D2AMD_POOL_FWD_MFMA=0 and once with =1, then compare:  python scripts/pool_fwd_mfma_check.py dump <file> | cmp <a> <b>"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def cases():
    import bench
    from detectron2_amd.modeling.poolers import ROIPooler
    from detectron2_amd.structures import Boxes
    dev = torch.device("cuda", 0)
    out = {}
    for dt in (torch.bfloat16, torch.float16):
        w = bench.Workload(dev, dt, "nhwc")
        out[f"box_{dt}"] = w.box_pooler([f.detach() for f in w.feats], w.box_lists)
        out[f"mask_{dt}"] = w.mask_pooler([f.detach() for f in w.feats], w.mask_lists)
        g = torch.Generator(device="cpu").manual_seed(5)
        n = 300
        ctr = torch.rand(n, 2, generator=g) * torch.tensor([1344., 800.])
        wh = torch.exp(torch.rand(n, 2, generator=g) * 7.5 - 0.5)  # 0.6 .. 1,100 px
        b = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
        b[:10] += 900.  # partly / fully outside
        b[10:14] = torch.tensor([[0., 0., 1344., 800.], [-50., -50., 30., 30.], [100., 100., 100., 100.], [5., 5., 4., 4.]])
        lists = [Boxes(b[: n // 2].to(dev)), Boxes(b[n // 2:].to(dev))]
        for res, sr in ((7, 2), (14, 2), (7, 0), (5, 3)):
            pl = ROIPooler(res, (1 / 4, 1 / 8, 1 / 16, 1 / 32), sr, "ROIAlignV2")
            out[f"rand_r{res}_s{sr}_{dt}"] = pl([f.detach() for f in w.feats], lists)
    torch.cuda.synchronize()
    return {k: v.float().cpu().numpy() for k, v in out.items()}

if sys.argv[1] == "dump":
    np.savez(sys.argv[2], **cases())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        d = np.abs(x - y)
        den = np.maximum(np.abs(x), 1e-3)
        ulp = 2.0 ** -8 if "bfloat16" in k else 2.0 ** -11
        bad = (d > 2 * ulp * den + 1e-6).sum()
        print(f"{k:34s} n={x.size:9d} differ={int((d > 0).sum()):8d} max rel {float((d / den).max()):.3e} beyond 2 ulp: {int(bad)}  nan {int(np.isnan(y).sum())}/{int(np.isnan(x).sum())}")
