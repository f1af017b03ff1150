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
if op == "connected_step":  # the headline's step itself (eager): its sampled ROI lists are what the paired gather sees there
    w2 = bench.Workload(dev, torch.bfloat16, layout)
    w2.connected = True
    for _ in range(N):
        bench.connected_step(w2)
elif op == "roi_align_chain_bwd":  # the step's backward: box and mask pooler of the same features, ONE pass, chained
    yb, ym = w.box_pooler(w.feats, w.box_lists), w.mask_pooler(w.feats, w.mask_lists)
    for _ in range(N):
        torch.autograd.grad([yb, ym], w.feats, [w.gbox, w.gmask], retain_graph=True)
elif op == "roi_align_pair_fwd":  # both poolers' forward in one launch (pool_pair)
    from detectron2_amd.modeling import pool_pair
    feats = [f.detach() for f in w.feats]
    for _ in range(N):
        pool_pair(w.box_pooler, w.mask_pooler, feats, w.box_lists, w.mask_lists)
elif op == "mask_targets":
    from detectron2_amd.structures import crop_and_resize_batch
    for _ in range(N):
        crop_and_resize_batch(w.gt_masks, [b.tensor for b in w.mask_lists], 28, w.fg_gt_index, w.crop_status)
elif op.startswith("roi_align_"):
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
elif op.startswith("dcn_"):  # dcn_fwd_res3 / dcn_bwd_res4 ...: DCNv2 at the R50 stage shapes, bf16, batch 2
    from detectron2_amd.layers import ModulatedDeformConv
    C, H, W = {"res3": (128, 100, 168), "res4": (256, 50, 84), "res5": (512, 25, 42)}[op.split("_")[2]]
    mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
    x = torch.randn(2, C, H, W, device=dev, dtype=torch.bfloat16)
    if layout == "nhwc":  # (the channels_last entry: column + dense GEMM path of round 5)
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    off = (torch.randn(2, 18, H, W, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
    msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(torch.bfloat16).requires_grad_(True)
    if "_fwd_" in op:
        for _ in range(N):
            mod(x.detach(), off.detach(), msk.detach())
    else:  # forward + backward per iteration (the backward consumes the column its forward saved); the summary drops
        g = None  # the forward's kernels by name
        for _ in range(N):
            y = mod(x, off, msk)
            g = torch.randn_like(y) if g is None else g  # (preserves channels_last)
            torch.autograd.grad([y], [x, off, msk, mod.weight], [g])
elif op == "paste_masks":  # SURVEY 8(d) paste micro: 100 masks of 28x28 -> 100 x 800 x 1333 bool
    from detectron2_amd.layers import paste_masks_in_image
    g = torch.Generator().manual_seed(7)
    masks = torch.rand(100, 28, 28, generator=g).to(dev)
    bx = bench.make_boxes(g, 100, 16, 600).to(dev)
    for _ in range(N):
        paste_masks_in_image(masks, bx, (800, 1333), 0.5)
elif op == "iou_rotated":  # SURVEY 8(d) rotated IoU: 16 x 268,569 (RRPN matching)
    import bench_extra
    from detectron2_amd.layers import pairwise_iou_rotated
    bench_extra.rotated_inputs(w)
    for _ in range(N):
        pairwise_iou_rotated(w.rot_gt[0], w.rot_anchors)
elif op == "retinanet_select":  # the dense-detector selection alone: 2 x 16.1 M logits, top 20,000 per level
    from detectron2_amd.modeling import dense_select_predictions
    an, lg, dl = bench.retina_inputs(dev, [0, 1])
    for _ in range(N):
        dense_select_predictions(an, lg, dl, 0.0, bench.RETINA_TOPK)
elif op in ("pool_rot_fwd", "pool_rot_bwd", "nms_rotated"):  # SURVEY 8(d) rotated rows at the rrpn_micro shapes
    import bench_extra
    from detectron2_amd.layers import batched_nms_rotated
    from detectron2_amd.modeling import ROIPooler
    bench_extra.rotated_inputs(w)
    if op == "nms_rotated":
        b, s, lv = w.rot_nms_in[0]
        for _ in range(N):
            batched_nms_rotated(b, s, lv, 0.7)
    else:
        pooler = ROIPooler(7, [1.0 / st for st in bench.STRIDES], 0, "ROIAlignRotated")
        rb = [bench_extra._RBoxes(r) for r in w.rot_rois]
        if op == "pool_rot_fwd":
            for _ in range(N):
                pooler([f.detach() for f in w.feats], rb)
        else:
            y = pooler(w.feats, rb)
            for _ in range(N):
                torch.autograd.grad([y], w.feats, [w.gbox], retain_graph=True)
elif op == "match_rpn":
    from detectron2_amd.modeling import Matcher
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    for _ in range(N):
        mt.match_boxes(w.gt[0], w.anchors)
torch.cuda.synchronize()
