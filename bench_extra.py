"""bench.py workloads added in round 4: the SURVEY 8(d) op rows that had no driver-visible line.

* maskrcnn_infer -- the Mask R-CNN R50-FPN INFERENCE hot path of one rank's 2 images: RPN test path (pre / post NMS
  top-k 1,000: proposal_generator/proposal_utils.py:67-135) -> box pooler 7x7 -> `fast_rcnn_inference` (roi_heads/
  fast_rcnn.py:118-170: clip, score > 0.05, per-class batched_nms at 0.5, top 100; in front of it predict_boxes /
  predict_probs, fast_rcnn.py:524-568, as ONE launch: fast_rcnn_predict) -> mask pooler 14x14 on the
  detections -> `mask_rcnn_inference` (mask_head.py:117-158) -> `paste_masks_in_image` 100 x 800x1333
  (postprocessing.py:60-68).  The box / mask heads' convolutions are inputs (synthetic scores / deltas / mask logits), as
  the backbone is for the training workload.  Roofline kernel: the paste (HBM write-bound, 106.6 MB per image).
* rrpn_micro -- the rotated operators on SURVEY 8(d)'s rotated inputs: `pairwise_iou_rotated` 16 ground-truth boxes x
  268,569 anchors per image (RRPN matching), `batched_nms_rotated` on 8,819 boxes / 5 levels per image at 0.7,
  ROIAlignRotated 7x7 forward + backward of 1,024 ROIs over the 4 FPN levels.  Roofline: the rotated IoU is VALU-bound
  (~400 flop / pair on 40 B of input): the line carries pairs/s beside the (small by construction) HBM fraction.

maskrcnn_infer runs as ONE HIP graph per step with one host read at its end (infer_step_device: fixed-shape intermediates
whose valid lengths stay on the device; the graph's detections and pasted masks are checked equal to the eager,
reference-style chain of infer_step before anything is timed; D2AMD_BENCH_INFER_EAGER=1 times that chain instead);
rrpn_micro is eager.  Both: barrier + synchronize bracketed timing, kernel durations from the library's launch-stream events
(d2amd_timing_*; for the graph: an eager pass right after the timed region)."""
import math
import os
import time

import torch

import bench as B


# ---------------------------------------------------------------------------------------------- maskrcnn_infer
INFER_PRE_NMS, INFER_POST_NMS, INFER_SCORE_THRESH, INFER_NMS, INFER_DETS = 1000, 1000, 0.05, 0.5, 100
INFER_LOOP = os.environ.get("D2AMD_BENCH_INFER_LOOP", "0") == "1"  # A/B: the reference's per-image box-head inference
INFER_TORCH_PREDICT = os.environ.get("D2AMD_BENCH_INFER_TORCH_PREDICT", "0") == "1"  # A/B: predict_boxes / predict_probs as torch ops
ORIG_H, ORIG_W = 800, 1333   # the image size detector_postprocess pastes at (BASELINE: 1333x800 inputs)


class _Inst:
    def __init__(self, classes):
        self.pred_classes = classes

    def __len__(self):
        return int(self.pred_classes.shape[0])


def infer_inputs(w, seed=1234):
    """Synthetic outputs of the box head and the mask head (inputs of the hot path): per image 1,000 x 81 class logits
    drawn so that ~5 % of the (proposal, class) pairs pass the 0.05 score threshold (<= 80 x 1,000
    candidates), class-specific deltas, and 100 x 80 x 28 x 28 mask logits."""
    gens = [B.image_generator(seed + 31, i) for i in w.image_ids]
    w.cls_logits = [(torch.randn(INFER_POST_NMS, 81, generator=g) * 2.0).to(w.dev) for g in gens]
    w.box_deltas = [(torch.randn(INFER_POST_NMS, 320, generator=g) * 0.1).to(w.dev) for g in gens]
    w.infer_mask_logits = [torch.randn(INFER_DETS, 80, 28, 28, generator=g).to(w.dtype).to(w.dev) for g in gens]
    # the heads produce ONE tensor for the batch (FastRCNNOutputLayers.forward / the mask head): concatenated once, here
    w.cls_logits_all, w.box_deltas_all = torch.cat(w.cls_logits), torch.cat(w.box_deltas)
    w.infer_mask_logits_all = torch.cat(w.infer_mask_logits)
    sx, sy = ORIG_W / B.IMG_W, ORIG_H / B.IMG_H   # detector_postprocess's scale (postprocessing.py:60-68): a constant of the model
    w.paste_scale = torch.tensor([sx, sy, sx, sy], device=w.dev)


def _apply_deltas(deltas, boxes, weights=(10.0, 10.0, 5.0, 5.0), clamp=math.log(1000.0 / 16)):
    """Box2BoxTransform.apply_deltas (modeling/box_regression.py:78-116), class-specific: deltas [R, 4K], boxes [R, 4].
    Plain torch ops: glue between two hot-path ops, exactly what the reference runs here."""
    wd = boxes[:, 2] - boxes[:, 0]
    ht = boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + 0.5 * wd
    cy = boxes[:, 1] + 0.5 * ht
    dx, dy = deltas[:, 0::4] / weights[0], deltas[:, 1::4] / weights[1]
    dw = (deltas[:, 2::4] / weights[2]).clamp(max=clamp)
    dh = (deltas[:, 3::4] / weights[3]).clamp(max=clamp)
    pcx, pcy = dx * wd[:, None] + cx[:, None], dy * ht[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * wd[:, None], torch.exp(dh) * ht[:, None]
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], dim=2)  # [R, K, 4]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """roi_heads/fast_rcnn.py:118-170 on this package's batched_nms: boxes [R, K, 4], scores [R, K + 1]."""
    from detectron2_amd.layers import batched_nms

    scores = scores[:, :-1]
    boxes = boxes.clone()
    boxes[..., 0::2].clamp_(0, image_shape[1])
    boxes[..., 1::2].clamp_(0, image_shape[0])
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()            # host sync, as in the reference (:150)
    boxes = boxes[filter_mask]
    scores = scores[filter_mask]
    keep = batched_nms(boxes, scores, filter_inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    return boxes[keep], scores[keep], filter_inds[keep, 1], int(filter_inds.shape[0])


def infer_step(w, run=None):
    """One pass of the inference hot path over the rank's images -> per image the pasted [n, 800, 1333] bool masks."""
    from detectron2_amd.layers import paste_masks_in_image
    from detectron2_amd.modeling import find_top_rpn_proposals_fused, mask_rcnn_inference
    from detectron2_amd.structures import Boxes

    run = run or (lambda name, fn: fn())
    n = w.n_img
    props = run("rpn_proposals_test", lambda: find_top_rpn_proposals_fused(
        w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, INFER_PRE_NMS, INFER_POST_NMS, 0.0, False))
    pboxes = []
    for p in props:  # the synthetic head outputs have 1,000 rows per image: pad short proposal lists (never at these sizes)
        t = p.proposal_boxes.tensor
        if t.shape[0] < INFER_POST_NMS:
            t = torch.cat([t, t.new_zeros(INFER_POST_NMS - t.shape[0], 4)])
        pboxes.append(t)
    box_feats = run("roi_align_box_fwd", lambda: w.box_pooler(w.feats_nograd, [Boxes(b) for b in pboxes]))

    def detect():
        # predict_probs / predict_boxes over the batch (fast_rcnn.py:FastRCNNOutputLayers: one softmax, one apply_deltas,
        # split per image), then fast_rcnn_inference: fused (d2amd_fast_rcnn_filter + ONE batched NMS: two host syncs per
        # batch) or -- D2AMD_BENCH_INFER_LOOP=1, the A/B -- the reference's per-image data flow on this package's ops
        rows = [int(p.shape[0]) for p in pboxes]
        if INFER_TORCH_PREDICT:  # A/B: the reference's ~45 elementwise launches
            probs = torch.softmax(w.cls_logits_all, dim=1)
            boxes = _apply_deltas(w.box_deltas_all, torch.cat(pboxes))
            probs, boxes = probs.split(rows), boxes.reshape(boxes.shape[0], -1).split(rows)
        else:
            from detectron2_amd.modeling import fast_rcnn_predict
            boxes, probs = fast_rcnn_predict(w.cls_logits_all, w.box_deltas_all, pboxes)
        if INFER_LOOP:
            return [fast_rcnn_inference_single_image(boxes[i].reshape(rows[i], -1, 4), probs[i], (B.IMG_H, B.IMG_W),
                                                     INFER_SCORE_THRESH, INFER_NMS, INFER_DETS) for i in range(n)]
        from detectron2_amd.modeling import fast_rcnn_inference_fused
        res, _rows = fast_rcnn_inference_fused(boxes, probs, [(B.IMG_H, B.IMG_W)] * n, INFER_SCORE_THRESH, INFER_NMS,
                                               INFER_DETS)
        return [(r.pred_boxes.tensor, r.scores, r.pred_classes, r.num_candidates) for r in res]

    dets = run("fast_rcnn_inference", detect)
    mask_feats = run("roi_align_mask_fwd", lambda: w.mask_pooler(w.feats_nograd, [Boxes(d[0]) for d in dets]))
    insts = [_Inst(d[2]) for d in dets]
    logits = torch.cat([w.infer_mask_logits[i][:len(insts[i])] for i in range(n)])
    run("mask_rcnn_inference", lambda: mask_rcnn_inference(logits, insts))
    # detector_postprocess (postprocessing.py:60-68): boxes scaled to the original image, masks pasted there
    sx, sy = ORIG_W / B.IMG_W, ORIG_H / B.IMG_H
    scale = torch.tensor([sx, sy, sx, sy], device=w.dev)
    pasted = run("paste_masks", lambda: [
        paste_masks_in_image(insts[i].pred_masks[:, 0], dets[i][0] * scale, (ORIG_H, ORIG_W), 0.5) for i in range(n)])
    return {"box_features": box_feats, "mask_features": mask_feats, "detections": dets, "masks": pasted}


def infer_step_device(w):
    """The same chain WITHOUT a host sync: fixed-shape intermediates whose valid lengths stay on the
    device -- the RPN test path's DeviceProposals (1,000 rows per image, counts read by the next kernels),
    fast_rcnn_inference_device (candidate windows, one batched NMS, 100 rows per image), the mask pooler / mask inference /
    paste on those 100 rows -- so the whole step is ONE HIP graph and the host reads once, at its end.  -> (finish, pasted):
    finish() = the read + the exact detection lists (as the synchronous path's), pasted[i] [100, H, W] valid up to them."""
    from detectron2_amd.layers import paste_masks_in_image
    from detectron2_amd.modeling import (fast_rcnn_inference_device, fast_rcnn_predict, find_top_rpn_proposals_fused,
                                         mask_rcnn_inference)
    from detectron2_amd.structures import Boxes

    n = w.n_img
    done = find_top_rpn_proposals_fused(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, INFER_PRE_NMS,
                                        INFER_POST_NMS, 0.0, False, defer=True, host_result=False)
    dp = done.device
    assert dp is not None, "the RPN NMS did not run as the batched device pipeline"
    pboxes = dp.pad_()   # rows behind the device-side proposal count <- a 1 x 1 box (one launch, no host read)
    box_feats = w.box_pooler(w.feats_nograd, [Boxes(b) for b in pboxes])
    # predict_boxes + predict_probs in one launch; rows behind the proposal count predict background
    boxes, probs = fast_rcnn_predict(w.cls_logits_all, w.box_deltas_all, pboxes, limits=dp.limits)
    dd = fast_rcnn_inference_device(boxes, probs, [(B.IMG_H, B.IMG_W)] * n, INFER_SCORE_THRESH, INFER_NMS, INFER_DETS,
                                    capacity=6144)
    mask_feats = w.mask_pooler(w.feats_nograd, [Boxes(b) for b in dd.boxes])
    insts = [_Inst(c) for c in dd.classes]
    mask_rcnn_inference(w.infer_mask_logits_all, insts)
    scale = w.paste_scale
    pasted = [paste_masks_in_image(insts[i].pred_masks[:, 0], dd.boxes[i] * scale, (ORIG_H, ORIG_W), 0.5) for i in range(n)]
    return dd.finish, pasted, (box_feats, mask_feats)


def bench_maskrcnn_infer(args, ctx):
    from detectron2_amd import _C as _dc
    from detectron2_amd.sharding import Stopwatch, global_image_ids

    dev, rank, world, dist = ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    ids = global_image_ids(B.IMAGES_PER_GPU, rank, world)
    w = B.Workload(dev, dtype, args.layout, image_ids=ids)
    w.feats_nograd = [f.detach() for f in w.feats]
    infer_inputs(w)
    # The chain as ONE HIP graph with one host read at its end (infer_step_device); D2AMD_BENCH_INFER_EAGER=1 (or a capture
    # error): the eager chain with the reference's host syncs (infer_step).  The graph's detections and pasted masks are
    # checked against the eager chain's before anything is timed.
    execution, step = None, None
    if os.environ.get("D2AMD_BENCH_INFER_EAGER") != "1" and not INFER_LOOP:
        try:
            with torch.no_grad():
                g, (fin, pasted, _feats) = B.GraphedStep._capture(lambda: infer_step_device(w))
                ref = infer_step(w)
                g.replay()
                dets, _rows = fin()
                for i in range(len(ids)):
                    m = len(dets[i].scores)
                    assert torch.equal(dets[i].pred_boxes.tensor, ref["detections"][i][0]) and torch.equal(dets[i].scores, ref["detections"][i][1])
                    assert torch.equal(pasted[i][:m], ref["masks"][i]), "pasted masks of the graph differ from the eager chain's"

            def step(w_, run=None):  # noqa: F811
                g.replay()
                d, _r = fin()
                return {"detections": [(x.pred_boxes.tensor, x.scores, x.pred_classes, x.num_candidates) for x in d],
                        "masks": [pasted[i][:len(d[i].scores)] for i in range(len(d))]}

            execution = "one HIP graph per step (RPN test path -> box pooler -> box-head inference -> mask pooler -> mask inference -> paste), ONE host read at its end; results checked equal to the eager chain's"
        except Exception as e:
            print(f"[bench] maskrcnn_infer: graph capture failed ({type(e).__name__}: {e}); eager", file=__import__("sys").stderr)
            torch.cuda.synchronize()
            step = None
    graphed = step is not None
    run_step = step if graphed else infer_step
    with torch.no_grad():
        for _ in range(args.warmup):
            out = run_step(w)
        knames = ["paste_masks", "pool_fwd_r7", "pool_fwd_r14", "nms_mask", "nms_reduce"]
        _dc.lib().d2amd_timing_select(",".join(knames).encode())
        sw = Stopwatch(dist, dev)
        sw.start()
        for _ in range(args.steps):
            out = run_step(w)
        elapsed = sw.stop()
        if graphed:  # (a replayed graph has no per-kernel events: an eager pass of the synchronous chain right after)
            for _ in range(min(args.steps, 10)):
                infer_step(w)
            torch.cuda.synchronize()
        ktimes = B.read_kernel_times(knames)
        _dc.lib().d2amd_timing_select(None)
        t = B.Timer()
        for _ in range(min(args.steps, 10)):
            infer_step(w, t.run)
        torch.cuda.synchronize()
    if rank != 0:
        return None
    n_img = len(ids)
    ndet = [int(m.shape[0]) for m in out["masks"]]
    s = w.esize
    paste_bytes = [nd * ORIG_H * ORIG_W + s * nd * 28 * 28 + 16 * nd for nd in ndet]  # N H W + s N M^2 + 16 N
    roof = {"bound": "hbm", "kernel": None, "achieved": None, "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
            "traffic": None}
    if "paste_masks" in ktimes:
        k_ms, k_n = ktimes["paste_masks"]
        kb = sum(paste_bytes) / n_img  # one launch = one image's detections
        roof = {"bound": "hbm", "kernel": "paste_region_kernel (+ the zero fill of the output it runs behind): one launch "
                                          "pastes one image's detections",
                "achieved": round(kb / 1e6 / k_ms, 1), "peak": B.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kb / 1e6 / k_ms / B.HBM_PEAK_GBS, 4), "traffic": B.pmc_traffic("paste_masks", "nhwc"),
                "traffic_source": B.pmc_source("paste_masks"), "alg_bytes_per_launch": int(kb),
                "ms_per_launch": round(k_ms, 4), "launches_timed": k_n,
                "alg_bytes_note": "SURVEY 8(d) paste: 1 B x N x H x W (bool out) + s N M^2 + 16 N, N = detections of the image",
                "timing": "HIP events recorded by the library on the kernel's launch stream, mean over the timed steps",
                "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
    res = {
        "metric": "img/s through the Mask R-CNN R50-FPN INFERENCE hot path (RPN test path, poolers, per-class NMS, mask "
                  "inference, paste at 800x1333), bs=2/GPU",
        "value": round(world * n_img * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "maskrcnn_r50fpn_inference_hotpath_bs2_800x1344 (SURVEY 8(d): paste / box-head NMS rows)",
                   "layout": args.layout, "global_batch": world * n_img, "proposals_per_image": INFER_POST_NMS,
                   "candidates_above_score_thresh": [d[3] for d in out["detections"]], "detections": ndet,
                   "paste_size": [ORIG_H, ORIG_W], "execution": execution or ("eager; box-head inference per image as the reference (A/B)" if INFER_LOOP else
                                 "eager; box-head inference fused over the batch (2 host syncs), the other syncs where the reference has them"),
                   "parallelism": f"dp{world}: images sharded, replicas only (inference, no collective)"},
        "roofline": roof,
        "ops": {k: {"ms_per_step": round(v, 4)} for k, v in t.totals_ms().items()},
    }
    for k, v in res["ops"].items():
        v["ms_per_step"] = round(v["ms_per_step"] / max(min(args.steps, 10), 1), 4)
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_infer(w, out)
    return res


def cpu_baseline_infer(w, out):
    """The REFERENCE's own Python for the paste (layers/mask_ops.py, via oracle/ref.py; kind "reference") + the C port
    for the NMS of image 0's candidates, on the host's cores.  Bounded sample: image 0's detections."""
    from oracle import ref

    if not ref.have_py():
        return None
    mo = ref.py_mask_ops()
    masks = torch.rand(out["masks"][0].shape[0], 28, 28)
    boxes = out["detections"][0][0].float().cpu()
    nthr = min(32, os.cpu_count() or 1)
    torch.set_num_threads(nthr)
    t = B._median_time(lambda: mo.paste_masks_in_image(masks, boxes, (ORIG_H, ORIG_W), 0.5), runs=3)
    return {"value": round(1.0 / t, 4), "unit": "img/s (paste only)", "cores": nthr, "kind": "reference",
            "sample": f"the reference's own layers/mask_ops.py:paste_masks_in_image (CPU path, torch threads = min(32, cores)) "
                      f"on image 0's {masks.shape[0]} detections at {ORIG_H}x{ORIG_W}: median of 3 = {t:.3f} s; the other "
                      f"stages of this workload have no reference CPU implementation outside torchvision"}


# ------------------------------------------------------------------------------------------------ rrpn_micro
def rotated_inputs(w, seed=1234):
    """the axis-aligned micro inputs with theta ~ U(-180, 180)."""
    gens = [B.image_generator(seed + 53, i) for i in w.image_ids]

    def rot(xyxy, g):
        cx, cy = (xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2
        wd, ht = xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]
        ang = torch.empty(xyxy.shape[0]).uniform_(-180, 180, generator=g)
        return torch.stack([cx, cy, wd, ht, ang], 1)

    w.rot_anchors = rot(w.anchors.cpu(), gens[0]).to(w.dev)                       # 268,569 x 5
    w.rot_gt = [rot(b.cpu(), g).to(w.dev) for b, g in zip(w.gt, gens)]            # 16 x 5 per image
    w.rot_nms_in = [(rot(b.cpu(), g).to(w.dev), s, lv) for (b, s, lv), g in zip(w.nms_in, gens)]  # 8,819 / 5 levels
    w.rot_rois = [rot(b.tensor.cpu(), g).to(w.dev) for b, g in zip(w.box_lists, gens)]            # 512 x 5 per image


class _RBoxes:
    """(cx, cy, w, h, angle) rows with the two members ROIPooler uses of structures.RotatedBoxes."""

    def __init__(self, t):
        self.tensor = t

    def __len__(self):
        return int(self.tensor.shape[0])

    def area(self):
        return self.tensor[:, 2] * self.tensor[:, 3]


def rrpn_step(w, run=None):
    from detectron2_amd.layers import batched_nms_rotated, pairwise_iou_rotated

    run = run or (lambda name, fn: fn())
    iou = run("iou_rotated", lambda: [pairwise_iou_rotated(g, w.rot_anchors) for g in w.rot_gt])
    keep = run("nms_rotated", lambda: [batched_nms_rotated(b, s, lv, 0.7) for b, s, lv in w.rot_nms_in])
    y = run("roi_align_rotated_fwd", lambda: w.rot_pooler(w.feats, [_RBoxes(r) for r in w.rot_rois]))
    run("roi_align_rotated_bwd", lambda: y.backward(w.gbox))
    for f in w.feats:
        f.grad = None
    return iou, keep, y


def bench_rrpn_micro(args, ctx):
    from detectron2_amd import _C as _dc
    from detectron2_amd.modeling import ROIPooler
    from detectron2_amd.sharding import Stopwatch, global_image_ids

    dev, rank, world, dist = ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    ids = global_image_ids(B.IMAGES_PER_GPU, rank, world)
    w = B.Workload(dev, dtype, args.layout, image_ids=ids, full=False)
    rotated_inputs(w)
    w.rot_pooler = ROIPooler(7, [1.0 / s for s in B.STRIDES], 0, "ROIAlignRotated")
    for _ in range(args.warmup):
        iou, keep, y = rrpn_step(w)
    knames = ["iou_rotated", "nms_mask", "nms_reduce", "roi_align_rot_fwd", "roi_align_rot_bwd", "pool_rot_fwd", "pool_rot_bwd"]
    _dc.lib().d2amd_timing_select(",".join(knames).encode())
    sw = Stopwatch(dist, dev)
    sw.start()
    for _ in range(args.steps):
        iou, keep, y = rrpn_step(w)
    elapsed = sw.stop()
    ktimes = B.read_kernel_times(knames)
    _dc.lib().d2amd_timing_select(None)
    t = B.Timer()
    for _ in range(min(args.steps, 10)):
        rrpn_step(w, t.run)
    torch.cuda.synchronize()
    if rank != 0:
        return None
    n_img = len(ids)
    n_gt, n_an = int(w.rot_gt[0].shape[0]), int(w.rot_anchors.shape[0])
    pairs_iou = n_gt * n_an                                     # per launch (= per image)
    iou_bytes = 20 * (n_gt + n_an) + 4 * n_gt * n_an            # 20 (N + M) + 4 N M
    per = (2000, 2000, 2000, 2000, 819)
    pairs_nms = sum(k * (k - 1) // 2 for k in per)
    roof = {"bound": "hbm", "kernel": None, "achieved": None, "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
            "traffic": None}
    if "iou_rotated" in ktimes:
        k_ms, k_n = ktimes["iou_rotated"]
        roof = {"bound": "hbm", "kernel": "box_iou_rotated_kernel (16 x 268,569 pairs per launch; VALU-bound: ~400 flop per "
                                          "pair on 40 B of input -- see pairs_per_s)",
                "achieved": round(iou_bytes / 1e6 / k_ms, 1), "peak": B.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(iou_bytes / 1e6 / k_ms / B.HBM_PEAK_GBS, 4), "traffic": B.pmc_traffic("iou_rotated", "nhwc"),
                "traffic_source": B.pmc_source("iou_rotated"), "alg_bytes_per_launch": int(iou_bytes),
                "ms_per_launch": round(k_ms, 4),
                "launches_timed": k_n, "alg_bytes_note": "SURVEY 8(d) box_iou_rotated: 20 (N + M) + 4 N M bytes",
                "pairs_per_s": round(pairs_iou / (k_ms / 1e3), 1),
                "timing": "HIP events recorded by the library on the kernel's launch stream, mean over the timed steps",
                "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
        if "nms_mask" in ktimes:
            roof["nms_rotated_pairs_per_s"] = round(pairs_nms / (ktimes["nms_mask"][0] / 1e3), 1)
    res = {
        "metric": "steps/s of the rotated-operator micro step (pairwise_iou_rotated 16 x 268,569, batched_nms_rotated "
                  "8,819 / 5 levels, ROIAlignRotated 7x7 fwd + bwd of 1,024 ROIs), 2 images/GPU",
        "value": round(world * args.steps / elapsed, 2), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "rrpn_micro_bs2_800x1344 (SURVEY 8(d) rotated rows)", "layout": args.layout,
                   "global_batch": world * n_img, "iou_pairs_per_image": pairs_iou, "nms_boxes_per_image": sum(per),
                   "nms_kept": [int(k.shape[0]) for k in keep], "rotated_rois": int(sum(r.shape[0] for r in w.rot_rois)),
                   "parallelism": f"dp{world}: images sharded, replicas only"},
        "roofline": roof,
        "ops": {k: {"ms_per_step": round(v / max(min(args.steps, 10), 1), 4)} for k, v in t.totals_ms().items()},
    }
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_rrpn(w)
    return res


def cpu_baseline_rrpn(w):
    """The COMPILED REFERENCE (oracle/_ref/libd2ref.so = the reference's own *_cpu.cpp, single-threaded by construction:
    ROIAlignRotated_cpu.cpp:216-217) on a bounded sample of the same inputs."""
    from oracle import ref

    if not ref.have_compiled():
        return None
    ops = ref.compiled()
    gt, an = w.rot_gt[0].cpu(), w.rot_anchors[:20000].cpu()
    t_iou = B._median_time(lambda: ops.box_iou_rotated(gt, an), runs=3)
    b, s, _ = w.rot_nms_in[0]
    b, s = b[:2000].cpu(), s[:2000].cpu()
    t_nms = B._median_time(lambda: ops.nms_rotated(b, s, 0.7), runs=3)
    x = w.feats[0][:1].detach().float().cpu().contiguous()
    r = w.rot_rois[0][:32].cpu()
    r6 = torch.cat([torch.zeros(32, 1), r], 1)
    t_ra = B._median_time(lambda: ops.roi_align_rotated_forward(x, r6, 0.25, 7, 7, 0), runs=3)
    pairs = gt.shape[0] * an.shape[0]
    return {"value": round(pairs / t_iou, 1), "unit": "rotated IoU pairs/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's own box_iou_rotated_cpu on 16 x 20,000 pairs: {t_iou:.3f} s; nms_rotated_cpu on the "
                      f"first 2,000 boxes of one level: {t_nms:.3f} s; ROIAlignRotated_forward_cpu, 32 ROIs x 256 ch on "
                      f"p2 of one image (fp32): {t_ra:.3f} s -- medians of 3 after a warm-up, 1 thread",
            "nms_rotated_2000_s": round(t_nms, 4), "roi_align_rotated_32rois_s": round(t_ra, 4),
            "host_cores_available": os.cpu_count()}
