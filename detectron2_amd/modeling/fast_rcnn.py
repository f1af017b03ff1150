"""Box-head inference (`fast_rcnn_inference`, roi_heads/fast_rcnn.py:118-170) for a whole batch: the fused counterpart
of the reference's per-image loop

    finite-row mask -> Boxes.clip -> scores > thresh -> nonzero [host sync] -> two boolean gathers -> batched_nms
    [host sync] -> keep[:topk] -> three index gathers

as ONE filter call for all images (d2amd_fast_rcnn_filter: candidates in torch.nonzero's order, no sync), one read of the
candidate counts, ONE batched NMS over all images (d2amd_nms_batched) and one read of its counts: two host syncs per
batch instead of two per image, and no torch op on the (row, class) matrix.  Same inputs as the reference's function
(`boxes` = predict_boxes, `scores` = predict_probs, per image), same results: the candidates, their order and the NMS
are the reference's bit for bit (tests/test_gpu_fast_rcnn.py)."""
import ctypes
import math
from typing import List, Sequence, Tuple

import torch

from .. import _C
from ..layers.nms import batched_nms_images
from ..structures import Boxes
from .dense_detector import Detections

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)  # box_regression.py:16

__all__ = ["fast_rcnn_inference_fused", "fast_rcnn_inference_device", "DeviceDetections", "fast_rcnn_predict"]


def fast_rcnn_inference_fused(boxes: Sequence[torch.Tensor], scores: Sequence[torch.Tensor],
                              image_shapes: Sequence[Tuple[int, int]], score_thresh: float, nms_thresh: float,
                              topk_per_image: int):
    """boxes[i] [R_i, K * 4] or [R_i, 4], scores[i] [R_i, K + 1] (HIP, any float dtype: evaluated in fp32 like the
    reference's fp32 head outputs), image_shapes[i] = (height, width).  -> (list of `Detections` with pred_boxes /
    scores / pred_classes, list of kept row indices `filter_inds[:, 0]`), as fast_rcnn.py:44-77 -- the indices count the
    rows that survive the finite-value check (the reference indexes boxes[valid_mask])."""
    n_img = len(boxes)
    assert n_img == len(scores) == len(image_shapes)
    if n_img == 0:
        return [], []
    _C.require_gpu(*boxes, *scores, op="fast_rcnn_inference")
    dev = boxes[0].device
    k_cls = int(scores[0].shape[1]) - 1
    kb = int(boxes[0].shape[1]) // 4
    assert k_cls >= 1 and kb in (1, k_cls), (k_cls, kb)
    bx = [b.detach().float().contiguous() for b in boxes]
    sc = [s.detach().float().contiguous() for s in scores]
    rows = [int(s.shape[0]) for s in sc]
    for b, s, r in zip(bx, sc, rows):
        assert b.shape == (r, kb * 4) and s.shape == (r, k_cls + 1), (b.shape, s.shape)
    base = [0]
    for r in rows:
        base.append(base[-1] + r * k_cls)
    cap = max(base[-1], 1)
    out_boxes = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((cap,), dtype=torch.float32, device=dev)
    out_classes = torch.empty((cap,), dtype=torch.int64, device=dev)
    out_rows = torch.empty((cap,), dtype=torch.int64, device=dev)
    counts = torch.zeros((n_img,), dtype=torch.int64, device=dev)
    L = _C.lib()
    with _C.on_device(dev):
        rows_c = (ctypes.c_int * n_img)(*rows)
        hw = (ctypes.c_int * (2 * n_img))(*[int(v) for s in image_shapes for v in s])
        ws_bytes = L.d2amd_fast_rcnn_filter_workspace_bytes(rows_c, n_img)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        ptrs = lambda ts: (ctypes.c_void_p * n_img)(*[t.data_ptr() for t in ts])
        _C.check(L.d2amd_fast_rcnn_filter(ptrs(bx), ptrs(sc), rows_c, n_img, k_cls, kb, hw, float(score_thresh),
                                          _C.ptr(out_boxes), _C.ptr(out_scores), _C.ptr(out_classes), _C.ptr(out_rows),
                                          _C.ptr(counts), _C.ptr(ws), ws_bytes, _C.stream()))
    cnt = counts.tolist()  # host sync 1 (the reference: one `nonzero` per image)
    cand = [(out_boxes[base[i]:base[i] + cnt[i]], out_scores[base[i]:base[i] + cnt[i]],
             out_classes[base[i]:base[i] + cnt[i]]) for i in range(n_img)]
    keeps = batched_nms_images(cand, nms_thresh)  # host sync 2 (the reference: one per image inside batched_nms)
    results, kept_rows = [], []
    for i, keep in enumerate(keeps):
        if topk_per_image >= 0:
            keep = keep[:topk_per_image]
        b, s, c = cand[i]
        det = Detections(tuple(image_shapes[i]), Boxes(b[keep]), s[keep], c[keep])
        det.num_candidates = cnt[i]  # (score > score_thresh pairs that went into the NMS)
        results.append(det)
        kept_rows.append(out_rows[base[i]:base[i] + cnt[i]][keep])
    return results, kept_rows


class DeviceDetections:
    """Box-head inference results of a batch with their counts left on the device: per image `boxes` [topk, 4], `scores`
    [topk], `classes` [topk] (int64), `rows` [topk] (kept row indices), valid up to `counts[i]` (int64 device tensor [N]);
    rows past the count hold a 1 x 1 box at the origin, score 0, class 0 -- harmless for the mask pooler / mask inference /
    paste that follow at fixed shape.  `finish()` performs the ONE host read of the step (kept counts + overflow flags,
    already copied to pinned memory behind the kernels) and returns exactly what fast_rcnn_inference_fused returns."""

    def __init__(self, boxes, scores, classes, rows, counts, finish):
        self.boxes, self.scores, self.classes, self.rows, self.counts, self.finish = boxes, scores, classes, rows, counts, finish


def fast_rcnn_inference_device(boxes: Sequence[torch.Tensor], scores: Sequence[torch.Tensor],
                               image_shapes: Sequence[Tuple[int, int]], score_thresh: float, nms_thresh: float,
                               topk_per_image: int, capacity: int = None) -> DeviceDetections:
    """fast_rcnn_inference (roi_heads/fast_rcnn.py:118-170) for the batch WITHOUT a host sync: every intermediate has a
    fixed shape, the counts are read by the next kernel on the device (what the training step does since round 3), so the
    whole inference chain can be captured in one HIP graph and the host reads once at its end.

      d2amd_fast_rcnn_filter (candidates in torch.nonzero's order + counts on the device)
      -> the first `capacity` candidate slots of every image, slots past the count parked (zero box, score -inf, a class
         of their own: they suppress nothing, are suppressed by nothing and sort last -- the RPN path's convention)
      -> ONE d2amd_nms_batched over all images -> the first topk kept rows, gathered at fixed shape.

    capacity: candidate slots per image (default: min(rows x classes, the batched NMS's limit of 12,288)).  An image with
    MORE candidates above `score_thresh` than that cannot be served at fixed shape: `finish()` sees its count and
    recomputes the batch through fast_rcnn_inference_fused (two syncs) -- results are always the reference's."""
    from ..layers import ops as _ops

    n_img = len(boxes)
    assert n_img == len(scores) == len(image_shapes) and n_img > 0 and topk_per_image > 0
    _C.require_gpu(*boxes, *scores, op="fast_rcnn_inference")
    dev = boxes[0].device
    k_cls = int(scores[0].shape[1]) - 1
    kb = int(boxes[0].shape[1]) // 4
    bx = [b.detach().float().contiguous() for b in boxes]
    sc = [s.detach().float().contiguous() for s in scores]
    rows = [int(s.shape[0]) for s in sc]
    base = [0]
    for r in rows:
        base.append(base[-1] + r * k_cls)
    L = _C.lib()
    if _ops._BATCH_MAX is None:
        _ops._BATCH_MAX = int(L.d2amd_nms_batched_max_boxes())
    cap = min(max(max(r * k_cls for r in rows), 1), _ops._BATCH_MAX, capacity or (1 << 30))
    # (a window never exceeds what ONE batched NMS takes -- `take` serves topk_per_image > window; ADVICE r05: raising
    # the window to topk_per_image pushed a huge top-k onto the per-image NMS path, whose result the gather below cannot read)
    cap = min(max(cap, topk_per_image), _ops._BATCH_MAX)
    total = max(base[-1], 1)
    out_boxes = torch.empty((total, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((total,), dtype=torch.float32, device=dev)
    out_classes = torch.empty((total,), dtype=torch.int64, device=dev)
    out_rows = torch.empty((total,), dtype=torch.int64, device=dev)
    # images without a slot (no rows) take no part in the NMS; its result rows and the candidate counts (int64) share
    # one zeroed buffer -- one fill, one mirror copy at the end
    win = [min(cap, rows[i] * k_cls) for i in range(n_img)]
    live_imgs = [i for i in range(n_img) if win[i] > 0]
    res = torch.zeros(8 * len(live_imgs) + 2 * n_img, dtype=torch.int32, device=dev)
    counts = res[8 * len(live_imgs):].view(torch.int64)
    with _C.on_device(dev):
        rows_c = (ctypes.c_int * n_img)(*rows)
        hw = (ctypes.c_int * (2 * n_img))(*[int(v) for s in image_shapes for v in s])
        ws_bytes = L.d2amd_fast_rcnn_filter_workspace_bytes(rows_c, n_img)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        ptrs = lambda ts: (ctypes.c_void_p * n_img)(*[t.data_ptr() for t in ts])
        _C.check(L.d2amd_fast_rcnn_filter(ptrs(bx), ptrs(sc), rows_c, n_img, k_cls, kb, hw, float(score_thresh),
                                          _C.ptr(out_boxes), _C.ptr(out_scores), _C.ptr(out_classes), _C.ptr(out_rows),
                                          _C.ptr(counts), _C.ptr(ws), ws_bytes, _C.stream()))
    with _C.on_device(dev):
        _C.check(L.d2amd_fast_rcnn_park(rows_c, n_img, k_cls, cap, _C.ptr(counts), _C.ptr(out_boxes), _C.ptr(out_scores),
                                        _C.ptr(out_classes), _C.stream()))
    # the windows themselves: views, no copies -- window_i = min(cap, rows_i x classes) slots, never beyond the image's
    # own slice
    cand = [(out_boxes[base[i]:base[i] + win[i]], out_scores[base[i]:base[i] + win[i]], out_classes[base[i]:base[i] + win[i]])
            for i in live_imgs]
    nms_done = _ops.nms_images(cand, nms_thresh, False, True, None, None, res, True) if live_imgs else None
    det_b = torch.empty((n_img, topk_per_image, 4), dtype=torch.float32, device=dev)
    det_s = torch.empty((n_img, topk_per_image), dtype=torch.float32, device=dev)
    det_c = torch.empty((n_img, topk_per_image), dtype=torch.int64, device=dev)
    det_r = torch.empty((n_img, topk_per_image), dtype=torch.int64, device=dev)
    n_valid = torch.empty((n_img,), dtype=torch.int64, device=dev)
    keeps = [None] * n_img
    for j, i in enumerate(live_imgs):
        keeps[i] = nms_done.keeps[j]
    with _C.on_device(dev):
        _C.check(L.d2amd_fast_rcnn_take(rows_c, n_img, k_cls, cap, int(topk_per_image),
                                        (ctypes.c_void_p * n_img)(*[None if k is None else k.data_ptr() for k in keeps]),
                                        res.data_ptr(), _C.ptr(out_boxes), _C.ptr(out_scores), _C.ptr(out_classes),
                                        _C.ptr(out_rows), _C.ptr(det_b), _C.ptr(det_s), _C.ptr(det_c), _C.ptr(det_r),
                                        _C.ptr(n_valid), _C.stream()))
    det_b, det_s, det_c, det_r = list(det_b), list(det_s), list(det_c), list(det_r)

    def finish():
        if nms_done is not None:
            kept_l, finite_l, tail = nms_done(with_finite=True)  # the one host read (pinned mirror behind the kernels)
        else:
            kept_l, finite_l, tail = [], [], res.tolist()
        cnts = [tail[2 * i] for i in range(n_img)]  # (low words of the int64 candidate counts)
        if any(c > w_ for c, w_ in zip(cnts, win)):  # more candidates than slots: the exact path
            return fast_rcnn_inference_fused(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image)
        results, kept_rows = [], []
        for i in range(n_img):
            m = 0
            if i in live_imgs:
                j = live_imgs.index(i)
                m = min(topk_per_image, len(kept_l[j]), finite_l[j])
            det = Detections(tuple(image_shapes[i]), Boxes(det_b[i][:m]), det_s[i][:m], det_c[i][:m])
            det.num_candidates = cnts[i]
            results.append(det)
            kept_rows.append(det_r[i][:m])
        return results, kept_rows

    return DeviceDetections(det_b, det_s, det_c, det_r, n_valid, finish)


def fast_rcnn_predict(scores: torch.Tensor, proposal_deltas: torch.Tensor, proposal_boxes: Sequence[torch.Tensor],
                      weights=(10.0, 10.0, 5.0, 5.0), scale_clamp: float = _DEFAULT_SCALE_CLAMP, use_sigmoid_ce: bool = False,
                      limits: Sequence[torch.Tensor] = None) -> Tuple[Tuple[torch.Tensor, ...], Tuple[torch.Tensor, ...]]:
    """`FastRCNNOutputLayers.predict_boxes` + `predict_probs` (roi_heads/fast_rcnn.py:524-568) for the batch in ONE launch:
    `scores` [R, K + 1] and `proposal_deltas` [R, K * 4] or [R, 4] are the box head's outputs for all images (R = sum of the
    images' proposal counts), `proposal_boxes[i]` [R_i, 4] the images' proposals, `weights` / `scale_clamp`
    Box2BoxTransform's.  -> (boxes, probs): per image [R_i, K_b * 4] fp32 (box_regression.py:88 decodes in fp32) and
    [R_i, K + 1] in the scores' dtype -- the two arguments of `fast_rcnn_inference`.  The reference runs ~45 elementwise
    launches here (apply_deltas' slices / divisions / exp / stack, the softmax); boxes are bit-identical to it, probabilities
    agree to the last ulp or two (reduction order of the softmax's sum).
    `limits[i]` (optional; `DeviceProposals.limits`): the proposals are a fixed-shape device-side list whose live length is
    min(limits[i][0], limits[i][2]) -- rows behind it predict background with probability 1 and zero boxes, no host read."""
    _C.require_gpu(scores, proposal_deltas, *proposal_boxes, op="fast_rcnn_predict")
    n_img = len(proposal_boxes)
    rows = [int(b.shape[0]) for b in proposal_boxes]
    r_tot = sum(rows)
    assert scores.dim() == 2 and proposal_deltas.dim() == 2 and scores.shape[0] == r_tot == proposal_deltas.shape[0], \
        (scores.shape, proposal_deltas.shape, rows)
    k_cls = int(scores.shape[1]) - 1
    kb = int(proposal_deltas.shape[1]) // 4
    assert proposal_deltas.shape[1] == kb * 4 and kb in (1, k_cls), (proposal_deltas.shape, k_cls)
    dev = scores.device
    out_dt = scores.dtype
    dt = out_dt if proposal_deltas.dtype == out_dt else torch.float32  # (mixed dtypes: both as fp32, probabilities cast back)
    sc = scores.detach().to(dt).contiguous()
    dl = proposal_deltas.detach().to(dt).contiguous()
    pb = [b.detach().float().contiguous() for b in proposal_boxes]
    boxes = torch.empty((r_tot, kb * 4), dtype=torch.float32, device=dev)
    probs = torch.empty((r_tot, k_cls + 1), dtype=dt, device=dev)
    if r_tot and n_img:
        ptrs = lambda ts: (ctypes.c_void_p * n_img)(*[(t.data_ptr() if t is not None and t.numel() else None) for t in ts])
        lim = None
        if limits is not None:
            assert len(limits) == n_img and all(l.dtype == torch.int64 and l.is_contiguous() and l.numel() >= 3 for l in limits)
            lim = ptrs(limits)
        wts = (ctypes.c_float * 4)(*[float(v) for v in weights])
        with _C.on_device(dev):
            _C.check(_C.lib().d2amd_fast_rcnn_predict(_C.ptr(sc), _C.ptr(dl), _C.dtype_code(sc), ptrs(pb), lim,
                                                      (ctypes.c_int * n_img)(*rows), n_img, k_cls, kb, wts, float(scale_clamp),
                                                      1 if use_sigmoid_ce else 0, _C.ptr(boxes), _C.ptr(probs), _C.stream()))
    return boxes.split(rows), probs.to(out_dt).split(rows)
