"""Per-op GPU time in isolation: each op is launched REP times back-to-back between two HIP events
(no host sync in between), so launch gaps overlap and the figure is device time per call."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.structures import pairwise_iou
from detectron2_amd.layers import paste_masks_in_image


def timeit(fn, rep=30, warm=3):
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
    layout = sys.argv[1] if len(sys.argv) > 1 else "nhwc"
    dev = torch.device("cuda", 0)
    w = bench.Workload(dev, torch.bfloat16, layout)
    alg = w.alg_bytes()
    res = {}
    res["pairwise_iou_rpn(1 img)"] = timeit(lambda: pairwise_iou(w.gt[0], w.anchors))
    for name, ops, rois, grads in (("box", w.box_ops, w.box_rois, w.gbox), ("mask", w.mask_ops, w.mask_rois, w.gmask)):
        for l in range(4):
            k = rois[l].shape[0]
            x = w.feats[l].detach()
            res[f"roi_fwd_{name}_p{l+2}(K={k})"] = timeit(lambda: ops[l](x, rois[l]))
            xg = w.feats[l]
            y = ops[l](xg, rois[l])
            res[f"roi_bwd_{name}_p{l+2}(K={k})"] = timeit(
                lambda: torch.autograd.grad(y, xg, grads[l], retain_graph=True))
    m = torch.rand(100, 28, 28, device=dev)
    xy = torch.rand(100, 2, device=dev) * torch.tensor([1333 * 0.8, 800 * 0.8], device=dev)
    wh = torch.rand(100, 2, device=dev) * torch.tensor([1333 * 0.4, 800 * 0.4], device=dev) + 4
    bx = torch.cat([xy, xy + wh], 1)
    res["paste_masks(100x800x1333)"] = timeit(lambda: paste_masks_in_image(m, bx, (800, 1333), 0.5))
    print(json.dumps({"layout": layout, "ms": {k: round(v, 4) for k, v in res.items()},
                      "alg_MB": {k: round(v / 1e6, 1) for k, v in alg.items()}}))


if __name__ == "__main__":
    main()
