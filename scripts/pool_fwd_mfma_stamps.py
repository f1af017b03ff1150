import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "box"
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
pooler, lists = (w.box_pooler, w.box_lists) if which == "box" else (w.mask_pooler, w.mask_lists)
for _ in range(3):
    pooler([f.detach() for f in w.feats], lists)
torch.cuda.synchronize()
os.environ["D2AMD_POOL_STAMPS"] = "/tmp/pool_stamps"
pooler([f.detach() for f in w.feats], lists)
torch.cuda.synchronize()
d = np.loadtxt("/tmp/pool_stamps.fwd", dtype=np.int64)
d = d[d[:, 1] > 0]
t0 = d[:, 1].min()
st, tb, lp, en = [(d[:, i] - t0) / 100.0 for i in (1, 2, 3, 4)]
ok = d[:, 3] > 0
mfma = os.environ.get("D2AMD_POOL_FWD_MFMA") == "1"
print(f"{which} fwd: {len(d)} workgroups ({ok.sum()} full), span {en.max():.1f} us; start p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f}")
if mfma:
    ch = (d[:, 5] >> 8) & 0xfff
    px = d[:, 5] >> 20
    for nm, a, b in (("tables", st, tb), ("chunks", tb, lp), ("epilogue", lp, en), ("total", st, en)):
        v = (b - a)[ok]
        print(f"  {nm:8s}: mean {v.mean():.2f} p50 {np.median(v):.2f} p90 {np.percentile(v, 90):.2f} max {v.max():.2f} us")
    print(f"  chunks per ROI mean {ch[ok].mean():.1f} max {ch[ok].max()}, footprint pixels mean {px[ok].mean():.0f} max {px[ok].max()}; per chunk {((lp - tb)[ok].sum() / ch[ok].sum()):.3f} us")
else:
    for nm, a, b in (("tables", st, tb), ("bins", tb, lp), ("total", st, lp)):
        v = (b - a)[ok]
        print(f"  {nm:8s}: mean {v.mean():.2f} p50 {np.median(v):.2f} p90 {np.percentile(v, 90):.2f} max {v.max():.2f} us")
