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
from typing import List, Sequence, Tuple

import torch

from .. import _C
from ..layers.nms import batched_nms_images
from ..structures import Boxes
from .dense_detector import Detections

__all__ = ["fast_rcnn_inference_fused"]


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
