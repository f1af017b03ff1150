"""Per-workgroup timeline of the DCN forward kernel (D2AMD_DCN_STAMPS): durations of the phases and the
number of workgroups in flight over time.   python scripts/dcn_stamps.py res3 [ablate]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.layers import ModulatedDeformConv

shapes = {"res3": (128, 100, 168), "res4": (256, 50, 84), "res5": (512, 25, 42)}
tag = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["D2AMD_DCN_ABLATE"] = sys.argv[2]
C, H, W = shapes[tag]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16)
off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16)
msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16)
for _ in range(3):
    mod(x, off, msk)
torch.cuda.synchronize()
os.environ["D2AMD_DCN_STAMPS"] = "/tmp/stamps.txt"
mod(x, off, msk)
torch.cuda.synchronize()
os.environ.pop("D2AMD_DCN_STAMPS")
d = np.loadtxt("/tmp/stamps.txt", dtype=np.int64)
d = d[d[:, 1] > 0]
t0 = d[:, 1].min()
st, tb, lp, en = [(d[:, i] - t0) / 100.0 for i in (1, 2, 3, 4)]  # us
print(f"{tag}: {len(d)} workgroups, span {en.max():.1f} us")
print(f"  start    : min {st.min():.1f} p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}")
for nm, a, b in (("tables", st, tb), ("loop", tb, lp), ("epilogue", lp, en), ("total", st, en)):
    v = b - a
    print(f"  {nm:9s}: mean {v.mean():.2f} p50 {np.median(v):.2f} p90 {np.percentile(v, 90):.2f} max {v.max():.2f} us")
order = np.argsort(st)
late = order[-16:]
print("  last-started wgs: start", np.round(st[late], 1), "dur", np.round((en - st)[late], 1))
