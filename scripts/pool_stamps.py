"""Per-workgroup timeline of the tile-gather backward (D2AMD_POOL_STAMPS).  python scripts/pool_stamps.py [box|mask]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "box"
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
if which == "pair":  # both poolers chained onto the same features: ONE paired gather
    from detectron2_amd.modeling import pool_pair
    y, grad = list(pool_pair(w.box_pooler, w.mask_pooler, w.feats, w.box_lists, w.mask_lists)), [w.gbox, w.gmask]
else:
    pooler, lists, g1 = (w.box_pooler, w.box_lists, w.gbox) if which == "box" else (w.mask_pooler, w.mask_lists, w.gmask)
    y, grad = [pooler(w.feats, lists)], [g1]
for _ in range(3):
    torch.autograd.grad(y, w.feats, grad, retain_graph=True)
torch.cuda.synchronize()
os.environ["D2AMD_POOL_STAMPS"] = "/tmp/pool_stamps"
torch.autograd.grad(y, w.feats, grad, retain_graph=True)
torch.cuda.synchronize()
os.environ.pop("D2AMD_POOL_STAMPS")
for ps, name in ((0, "fine levels"), (1, "coarse levels")):
    if os.path.getsize(f"/tmp/pool_stamps.pass{ps}") == 0:
        continue  # single-launch (LDS-staged) backward: everything is in pass 0
    d = np.loadtxt(f"/tmp/pool_stamps.pass{ps}", dtype=np.int64)
    d = d[d[:, 1] > 0]
    t0 = d[:, 1].min()
    st, ls, lp, en = [(d[:, i] - t0) / 100.0 for i in (1, 2, 3, 4)]
    ls = np.where(d[:, 2] > 0, ls, st)
    kk = (d[:, 5] >> 12) & 0xfffff  # (K-concatenated gather: bins of the tile's windows = k's of its contraction)
    n = d[:, 5] & 0xfff  # (bits 32+: the XCD whose workgroup ran the tile; the row index & 7 = the queue's XCD)
    stolen = ((d[:, 5] >> 32) & 7) != (d[:, 0] & 7)
    print(f'  tiles taken from another XCD\'s queue: {int(stolen.sum())}')
    print(f"{which} {name}: {len(d)} workgroups, span {en.max():.1f} us; ROIs/tile mean {n.mean():.2f} max {n.max()} zero {np.mean(n == 0):.2f}")
    print(f"  start p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}")
    for nm, a, b in (("scan", st, ls), ("rois", ls, lp), ("write", lp, en), ("total", st, en)):
        v = b - a
        print(f"  {nm:6s}: mean {v.mean():.2f} p50 {np.median(v):.2f} p90 {np.percentile(v, 90):.2f} max {v.max():.2f} us")
    part = (d[:, 5] >> 56) & 1
    top = np.argsort(-(en - st))[:8]
    print("  longest tiles (us, start, ROIs, k's, part of a split list, logical id): " + ", ".join(f"{(en - st)[i]:.1f}@{st[i]:.1f} n={n[i]} k={kk[i]} p={part[i]} id={d[i, 0]}" for i in top))
    if kk.sum() > 0:
        A = np.stack([np.ones(len(d)), n.astype(float), kk.astype(float)], 1)
        coef, *_ = np.linalg.lstsq(A, (en - st), rcond=None)
        print(f"  total us per tile ~ {coef[0]:.2f} + {coef[1]:.3f} n + {coef[2]:.4f} k  (k mean {kk.mean():.1f}, sum {kk.sum()})")
        for lo_, hi_ in ((0, 16), (16, 48), (48, 96), (96, 192), (192, 384), (384, 10 ** 6)):
            m = (kk > lo_) & (kk <= hi_)
            if m.any():
                print(f"    k in ({lo_}, {hi_}]: {m.sum()} tiles, total mean {(en - st)[m].mean():.2f} us, n mean {n[m].mean():.1f}")
    print(f"  parts of split lists: {int(part.sum())}")
    for k in (0, 1, 2, 4, 8, 16, 32):
        m = n == k
        if m.any():
            print(f"  tiles with {k} ROIs: {m.sum()}, total mean {(en - st)[m].mean():.2f} us, rois-phase mean {(lp - ls)[m].mean():.2f}")
    # chains of the persistent workgroups (bits 32..55 of the last stamp = blockIdx.x): busy time, gaps between tiles
    wg = (d[:, 5] >> 32) & 0xffffff
    if len(np.unique(wg)) > 8:
        busy, gaps, ends, cnts = [], [], [], []
        for b in np.unique(wg):
            m = wg == b
            o = np.argsort(st[m])
            s_, e_ = st[m][o], en[m][o]
            busy.append((e_ - s_).sum()); ends.append(e_.max()); cnts.append(m.sum())
            gaps += list(s_[1:] - e_[:-1])
        busy, ends, gaps, cnts = map(np.array, (busy, ends, gaps, cnts))
        print(f"  per workgroup ({len(busy)}): tiles mean {cnts.mean():.2f} max {cnts.max()}; busy mean {busy.mean():.1f} p90 {np.percentile(busy, 90):.1f} max {busy.max():.1f} us; "
              f"last end mean {ends.mean():.1f} p10 {np.percentile(ends, 10):.1f} max {ends.max():.1f}; gap between tiles mean {gaps.mean():.2f} p90 {np.percentile(gaps, 90):.2f} max {gaps.max():.2f} us")
        for q in range(8):
            m = (wg & 7) == q
            print(f"    XCD {q}: {m.sum()} tiles, rois {n[m].sum()}, busy sum {(en - st)[m].sum():.0f} us, first start {st[m].min():.1f} last end {en[m].max():.1f}")
