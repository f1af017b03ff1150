"""Per-op GPU time in isolation: each op is launched REP times back-to-back between two HIP events
(no host sync in between), so launch gaps overlap and the figure is device time per call.
Covers every op of SURVEY.md 8(a) at the 8(d) micro-benchmark shapes.

    python scripts/microbench.py [nhwc|nchw] [--quick]
"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.layers import (ModulatedDeformConv, DeformConv, batched_nms, nms, paste_masks_in_image,
                                   pairwise_iou_rotated, nms_rotated)
from detectron2_amd.structures import pairwise_iou

HBM, MFMA_BF16 = 8000.0, 2500.0  # GB/s, TFLOP/s (MI355X_MICROARCH.md)


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


def nms_inputs(gen, n, ncls, dev):
    s = torch.exp(torch.empty(n).uniform_(math.log(8), math.log(400), generator=gen))
    ar = torch.exp(torch.empty(n).uniform_(math.log(0.5), math.log(2.0), generator=gen))
    w, h = s * ar.sqrt(), s / ar.sqrt()
    cx = torch.empty(n).uniform_(0, 1344, generator=gen)
    cy = torch.empty(n).uniform_(0, 800, generator=gen)
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    sc = torch.rand(n, generator=gen) + torch.arange(n) * 1e-9
    idx = torch.randint(0, ncls, (n,), generator=gen)
    return b.to(dev), sc.to(dev), idx.to(dev)


def main():
    layout = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "nhwc"
    quick = "--quick" in sys.argv
    dev = torch.device("cuda", 0)
    w = bench.Workload(dev, torch.bfloat16, layout)
    alg = w.alg_bytes()
    res, extra = {}, {}

    def rec(name, ms, bytes_=None, flops=None):
        res[name] = round(ms, 4)
        if bytes_:
            extra[name] = {"alg_MB": round(bytes_ / 1e6, 2), "GBps": round(bytes_ / 1e6 / ms, 1),
                           "frac_hbm": round(bytes_ / 1e6 / ms / HBM, 4)}
        if flops:
            extra[name] = {"GFLOP": round(flops / 1e9, 2), "TFLOPs": round(flops / 1e9 / ms, 1),
                           "frac_mfma_bf16": round(flops / 1e9 / ms / MFMA_BF16, 4)}

    rec("pairwise_iou_rpn(16x268569)", timeit(lambda: pairwise_iou(w.gt[0], w.anchors)), 16 * (16 + 268569) + 4 * 16 * 268569)
    rec("pairwise_iou_roi(16x1016)", timeit(lambda: pairwise_iou(w.gt[0], w.props[0])))
    # fused IoU + Matcher (f3), batch NMS in one call and the fused RPN proposal path (f2)
    from detectron2_amd.layers import batched_nms_images
    from detectron2_amd.modeling import Matcher, find_top_rpn_proposals_fused
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    rec("match_boxes_rpn(16x268569,fused iou+matcher)", timeit(lambda: mt.match_boxes(w.gt[0], w.anchors)),
        2 * 16 * 268569 + 9 * 268569)
    rec("batched_nms_images(2x8819)", timeit(lambda: batched_nms_images(w.nms_in, 0.7)))
    gen0 = torch.Generator().manual_seed(11)
    sizes = [201600, 50400, 12600, 3150, 819]
    A = [bench.make_boxes(gen0, a, 16, 512).to(dev) for a in sizes]
    Lg = [(torch.randn(2, a, generator=gen0) + torch.arange(a) * 1e-7).to(dev) for a in sizes]
    Dl = [(torch.randn(2, a, 4, generator=gen0) * 0.2).to(dev) for a in sizes]
    rec("find_top_rpn_proposals_fused(2 img,268569 anchors,2000/1000)",
        timeit(lambda: find_top_rpn_proposals_fused(A, Lg, Dl, [(800, 1344)] * 2, 0.7, 2000, 1000, 0.0, True), rep=10))
    if not quick:  # BASELINE configs[3]: RetinaNet R50-FPN inference selection, 2 x 16.1 M class scores
        from detectron2_amd.modeling import dense_detector_inference_fused, dense_select_predictions
        rs = [9 * 16800, 9 * 4200, 9 * 1050, 9 * 273, 9 * 77]
        RA = [bench.make_boxes(gen0, a, 16, 512).to(dev) for a in rs]
        RL = [(torch.randn(2, a, 80, generator=gen0) * 1.2 - 4.6).to(dev) for a in rs]
        RD = [(torch.randn(2, a, 4, generator=gen0) * 0.2).to(dev) for a in rs]
        nbytes = sum(x.numel() for x in RL) * 4
        rec("dense_select_predictions(retinanet 2 img,16.1M scores/img,thr .05,top 1000)",
            timeit(lambda: dense_select_predictions(RA, RL, RD, 0.05, 1000), rep=10), 4 * nbytes)
        rec("dense_detector_inference_fused(retinanet 2 img,+nms .5,top 100)",
            timeit(lambda: dense_detector_inference_fused(RA, RL, RD, [(800, 1344)] * 2, 0.05, 1000, 0.5, 100), rep=10))
        del RL, RD
    # mask-head glue: loss forward + backward and inference select at the Mask R-CNN shapes
    from detectron2_amd.modeling import mask_rcnn_inference, mask_rcnn_loss_from_targets
    ml = (torch.randn(256, 80, 28, 28, generator=gen0)).to(dev).bfloat16().requires_grad_(True)
    mc = torch.randint(0, 80, (256,), generator=gen0).to(dev)
    mg = (torch.rand(256, 28, 28, generator=gen0) < 0.4).to(dev)
    rec("mask_rcnn_loss forward(256x80x28x28 bf16)", timeit(lambda: mask_rcnn_loss_from_targets(ml, mc, mg)),
        256 * 784 * 3)
    from detectron2_amd import _C as _dc
    _g, _one, _t8, _x = torch.empty_like(ml), torch.ones((), device=dev), mg.view(torch.uint8), ml.detach()
    rec("mask_rcnn_loss backward kernel(256x80x28x28 bf16)",
        timeit(lambda: _dc.lib().d2amd_mask_rcnn_loss_backward(_dc.ptr(_x), _dc.ptr(mc), _dc.ptr(_t8), _dc.ptr(_one), 256, 80,
                                                               784, _dc.dtype_code(_x), _dc.ptr(_g), _dc.stream())),
        2 * ml.numel() + 3 * 256 * 784)
    class _I:
        pred_classes = mc
        def __len__(self): return 256
    rec("mask_rcnn_inference(256x80x28x28 bf16)", timeit(lambda: mask_rcnn_inference(ml.detach(), [_I()])))
    for name, pooler, lists, grad in (("box7", w.box_pooler, w.box_lists, w.gbox),
                                      ("mask14", w.mask_pooler, w.mask_lists, w.gmask)):
        key = "roi_align_box" if name == "box7" else "roi_align_mask"
        rec(f"pooler_fwd_{name}", timeit(lambda: pooler([f.detach() for f in w.feats], lists)), alg[key + "_fwd"])
        y = pooler(w.feats, lists)
        rec(f"pooler_bwd_{name}", timeit(lambda: torch.autograd.grad([y], w.feats, [grad], retain_graph=True)),
            alg[key + "_bwd"])
        del y
    gen = torch.Generator().manual_seed(7)
    b, s, lv = w.nms_in[0]
    rec("batched_nms_rpn(8819,5cls,.7)", timeit(lambda: batched_nms(b, s, lv, 0.7)))
    b2, s2, i2 = nms_inputs(gen, 20000, 80, dev)
    rec("batched_nms_boxhead(20000,80cls,.5)", timeit(lambda: batched_nms(b2, s2, i2, 0.5), rep=10))
    b3, s3, i3 = nms_inputs(gen, 100000, 80, dev)
    rec("batched_nms_retinanet(100000,80cls,.5)", timeit(lambda: batched_nms(b3, s3, i3, 0.5), rep=5))
    rec("nms(4096,.5)", timeit(lambda: nms(b2[:4096], s2[:4096], 0.5)))
    m = torch.rand(100, 28, 28, device=dev)
    xy = torch.rand(100, 2, device=dev) * torch.tensor([1333 * 0.8, 800 * 0.8], device=dev)
    wh = torch.rand(100, 2, device=dev) * torch.tensor([1333 * 0.4, 800 * 0.4], device=dev) + 4
    bx = torch.cat([xy, xy + wh], 1)
    rec("paste_masks(100x800x1333)", timeit(lambda: paste_masks_in_image(m, bx, (800, 1333), 0.5)),
        100 * 800 * 1333 + 4 * 100 * 28 * 28 + 1600)
    if not quick:
        rb = torch.cat([(b2[:2000, :2] + b2[:2000, 2:]) / 2, b2[:2000, 2:] - b2[:2000, :2],
                        torch.rand(2000, 1, device=dev) * 360 - 180], 1)
        rec("pairwise_iou_rotated(64x2000)", timeit(lambda: pairwise_iou_rotated(rb[:64], rb), rep=5))
        rec("nms_rotated(2000,.5)", timeit(lambda: nms_rotated(rb, s2[:2000], 0.5), rep=5))
        # DCNv2, the three R50 stages of SURVEY 8(a) a5/a6, bf16
        for tag, (C, H, W) in (("res3", (128, 100, 168)), ("res4", (256, 50, 84)), ("res5", (512, 25, 42))):
            mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
            x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16, requires_grad=True)
            off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
            msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16).requires_grad_(True)
            flops = 2.0 * C * C * 9 * 2 * H * W
            rec(f"dcnv2_fwd_{tag}", timeit(lambda: mod(x.detach(), off.detach(), msk.detach()), rep=10), flops=flops)
            y = mod(x, off, msk)
            g = torch.randn_like(y)
            rec(f"dcnv2_bwd_{tag}", timeit(lambda: torch.autograd.grad(
                [y], [x, off, msk, mod.weight], [g], retain_graph=True), rep=10), flops=2 * flops)
            del y
    print(json.dumps({"layout": layout, "ms": res, "derived": extra}))


if __name__ == "__main__":
    main()
