"""Per-workgroup timeline of tk_gather_kernel (D2AMD_TOPK_STAMPS) on the bench's RetinaNet logits.
    python scripts/topk_stamps.py        (on a GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
os.environ["D2AMD_TOPK_STAMPS"] = "/tmp/topk_stamps"
import bench  # noqa: E402
from detectron2_amd.modeling import dense_select_predictions  # noqa: E402

dev = torch.device("cuda")
anchors, logits, deltas = bench.retina_inputs(dev, [0, 1])
for _ in range(3):
    dense_select_predictions(anchors, logits, deltas, 0.0, bench.RETINA_TOPK)
torch.cuda.synchronize()
a = np.loadtxt("/tmp/topk_stamps")
t0 = a[:, 1].min()
st = (a[:, 1:] - t0) / 100.0  # us (100 MHz)
st[a[:, 1:] == 0] = np.nan
seg = (a[:, 0] // (a[:, 0].max() // 10 + 1)).astype(int)
print("workgroups", len(a), "span us", np.nanmax(st))
names = ["start", "loaded+scanned", "classified", "reserved", "copied", "dense"]
for k in range(6):
    col = st[:, k]
    if np.isfinite(col).any():
        print("%-16s n=%5d  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (names[k], np.isfinite(col).sum(), np.nanmin(col), np.nanmedian(col), np.nanpercentile(col, 90), np.nanmax(col)))
for k, (i, j) in {"load+scan": (0, 1), "classify": (1, 2), "reserve": (2, 3), "copy": (3, 4)}.items():
    d = st[:, j] - st[:, i]
    print("%-10s p50 %6.2f  p90 %6.2f  max %6.2f us" % (k, np.nanmedian(d), np.nanpercentile(d, 90), np.nanmax(d)))
# start times histogram: how many workgroups started in each 5 us bucket
h, _ = np.histogram(st[:, 0], bins=np.arange(0, np.nanmax(st) + 5, 5))
print("starts per 5 us:", h.tolist())
end = np.nanmax(st, axis=1)
h, _ = np.histogram(end, bins=np.arange(0, np.nanmax(st) + 5, 5))
print("ends per 5 us:  ", h.tolist())
