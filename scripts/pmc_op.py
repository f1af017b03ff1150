"""Run ONE op of the bench step N times (for `rocprofv3 --pmc <COUNTER>` passes: the per-kernel counter
rows of the run are summed and divided by N by scripts/pmc_summary.py).
    python scripts/pmc_op.py <op> <layout> <N>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.layers import batched_nms
from detectron2_amd.structures import pairwise_iou

op, layout, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
dev = torch.device("cuda", 0)
w = bench.Workload(dev, torch.bfloat16, layout)
torch.cuda.synchronize()
if op.startswith("roi_align_"):
    box = "box" in op
    pooler, lists, grad = (w.box_pooler, w.box_lists, w.gbox) if box else (w.mask_pooler, w.mask_lists, w.gmask)
    if op.endswith("_fwd"):
        for _ in range(N):
            pooler([f.detach() for f in w.feats], lists)
    else:
        y = pooler(w.feats, lists)   # one forward (its kernels are excluded by name in the summary)
        for _ in range(N):
            torch.autograd.grad([y], w.feats, [grad], retain_graph=True)
elif op == "pairwise_iou_rpn":
    for _ in range(N):
        pairwise_iou(w.gt[0], w.anchors)
elif op == "batched_nms_rpn":
    b, s, lv = w.nms_in[0]
    for _ in range(N):
        batched_nms(b, s, lv, 0.7)
torch.cuda.synchronize()
