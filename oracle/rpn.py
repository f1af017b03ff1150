"""CPU restatement (numpy, fp32) of the RPN proposal path -- TEST INFRASTRUCTURE ONLY.

  apply_deltas            <- detectron2/modeling/box_regression.py:71-116 (Box2BoxTransform.apply_deltas)
  find_top_rpn_proposals  <- detectron2/modeling/proposal_generator/proposal_utils.py:22-135, with
                             `batched_nms` = oracle.batched_nms (the torchvision restatement)
Pinned against the reference's own functions (loaded with import stubs by oracle/ref.py::py_box_regression /
py_proposal_utils) through tests/golden/rpn_proposals.npz.  Decoded coordinates agree with the reference to
~1 ulp of exp() (torch.exp vs numpy.exp), everything else exactly."""
import math

import numpy as np

from . import batched_nms

DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


def apply_deltas(deltas, boxes, weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=DEFAULT_SCALE_CLAMP):
    f = np.float32
    deltas = np.asarray(deltas, f)
    boxes = np.asarray(boxes, f)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + f(0.5) * widths
    ctr_y = boxes[:, 1] + f(0.5) * heights
    wx, wy, ww, wh = [f(w) for w in weights]
    dx, dy = deltas[:, 0] / wx, deltas[:, 1] / wy
    dw, dh = deltas[:, 2] / ww, deltas[:, 3] / wh
    with np.errstate(invalid="ignore", over="ignore"):
        dw = np.where(np.isnan(dw), dw, np.minimum(dw, f(scale_clamp)))
        dh = np.where(np.isnan(dh), dh, np.minimum(dh, f(scale_clamp)))
        pcx = dx * widths + ctr_x
        pcy = dy * heights + ctr_y
        pw = np.exp(dw).astype(f) * widths
        ph = np.exp(dh).astype(f) * heights
        out = np.stack([pcx - f(0.5) * pw, pcy - f(0.5) * ph, pcx + f(0.5) * pw, pcy + f(0.5) * ph], 1)
    return out.astype(f)


def topk_desc(logits, k):
    """indices of the k largest, descending, ties -> lower index first (stable)."""
    order = np.argsort(-logits.astype(np.float64), kind="stable")
    return order[:k]


def select(anchors, logits, deltas, image_sizes, pre_nms_topk, min_box_size, weights=(1.0, 1.0, 1.0, 1.0),
           scale_clamp=DEFAULT_SCALE_CLAMP):
    """-> per image: (boxes [K,4] clipped, scores [K], valid [K] bool, level [K], anchor index in level [K])."""
    n = logits[0].shape[0]
    res = []
    for i in range(n):
        bs, ss, vs, ls, ids = [], [], [], [], []
        h, w = image_sizes[i]
        for l, (a, lg, dl) in enumerate(zip(anchors, logits, deltas)):
            k = min(a.shape[0], pre_nms_topk)
            idx = topk_desc(lg[i], k)
            b = apply_deltas(dl[i][idx], a[idx], weights, scale_clamp)
            s = lg[i][idx].astype(np.float32)
            fin = np.isfinite(b).all(1) & np.isfinite(s)
            with np.errstate(invalid="ignore"):
                c = b.copy()
                c[:, 0::2] = np.clip(c[:, 0::2], 0, np.float32(w))
                c[:, 1::2] = np.clip(c[:, 1::2], 0, np.float32(h))
                ok = fin & ((c[:, 2] - c[:, 0]) > np.float32(min_box_size)) & ((c[:, 3] - c[:, 1]) > np.float32(min_box_size))
            bs.append(c); ss.append(s); vs.append(ok); ls.append(np.full(k, l, np.int64)); ids.append(idx)
        res.append(tuple(np.concatenate(v) for v in (bs, ss, vs, ls, ids)))
    return res


def find_top_rpn_proposals(anchors, logits, deltas, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk,
                           min_box_size, weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=DEFAULT_SCALE_CLAMP,
                           selected=None):
    """-> per image (proposal_boxes [P,4], objectness_logits [P]).  `selected` (the output of select(), or the
    device's decoded boxes) lets the caller run the NMS part on given decode results."""
    sel = selected if selected is not None else select(anchors, logits, deltas, image_sizes, pre_nms_topk,
                                                       min_box_size, weights, scale_clamp)
    out = []
    for b, s, ok, lv, _ in sel:
        b, s, lv = b[ok], s[ok], lv[ok]
        keep = batched_nms(b, s, lv, nms_thresh)[:post_nms_topk]
        out.append((b[keep], s[keep]))
    return out
