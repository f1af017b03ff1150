"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this): numpy
restatement of the reference's box-head inference, detectron2/modeling/roi_heads/fast_rcnn.py:118-170
(`fast_rcnn_inference_single_image`), on the C port of batched_nms (oracle.batched_nms: torchvision's coordinate trick
and greedy NMS, pinned to the reference's known answers in tests/test_oracle_golden.py).  Parity pin: the function is
checked against the reference's OWN `fast_rcnn_inference_single_image` source executed on the CPU
(tests/test_oracle_golden.py::test_fast_rcnn_inference_vs_reference_source, needs /root/reference)."""
import numpy as np

import oracle


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """boxes [R, K * 4] or [R, 4] fp32, scores [R, K + 1] fp32, image_shape (h, w) ->
    (boxes [n, 4], scores [n], classes [n] int64, rows [n] int64: indices among the rows that were not dropped)."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    valid = np.isfinite(boxes).all(axis=1) & np.isfinite(scores).all(axis=1)  # :134-137
    boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]                                                   # :139
    kb = boxes.shape[1] // 4
    b = boxes.reshape(-1, 4).copy()                                           # :142-144 Boxes.clip
    h, w = image_shape
    b[:, 0] = np.clip(b[:, 0], 0, w); b[:, 1] = np.clip(b[:, 1], 0, h)
    b[:, 2] = np.clip(b[:, 2], 0, w); b[:, 3] = np.clip(b[:, 3], 0, h)
    b = b.reshape(-1, kb, 4)
    mask = scores > np.float32(score_thresh)                                  # :148
    inds = np.argwhere(mask)                                                  # :151 nonzero: row-major
    cb = b[inds[:, 0], 0] if kb == 1 else b[mask]                             # :152-155
    cs = scores[mask]
    keep = oracle.batched_nms(cb, cs, inds[:, 1].astype(np.int64), nms_thresh)  # :159
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    # (:170 returns filter_inds[:, 0]: indices into boxes[valid_mask], not into the caller's rows)
    return cb[keep], cs[keep], inds[keep, 1].astype(np.int64), inds[keep, 0].astype(np.int64)
