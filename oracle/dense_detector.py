"""CPU restatement (numpy, fp32) of the dense-detector (RetinaNet) inference path -- TEST INFRASTRUCTURE ONLY.

  decode_per_level          <- detectron2/modeling/meta_arch/dense_detector.py:186-232 (_decode_per_level_predictions:
                               `scores > score_thresh`, nonzero, topk, apply_deltas of the selected anchors)
  decode_multi_level        <- dense_detector.py:234-260 (per level, concatenated)
  inference_single_image    <- detectron2/modeling/meta_arch/retinanet.py:275-309 (decode, batched_nms, top
                               max_detections), with `batched_nms` = oracle.batched_nms (the torchvision restatement)
Scores are sigmoid(logit) in fp32 as retinanet.py:267 computes them.

RANKING RULE (shared with the HIP path, csrc/topk.hip).  The reference ranks the fp32 scores with `torch.topk`, whose
order between equal scores is unspecified, and whose input depends on the exp() of the build it runs on (a 1-ulp
difference reorders near-ties).  sigmoid is monotone, so both sides rank the LOGIT instead: logit descending, equal
logits towards the lower flattened index a*K + c (IEEE total order: +0 ahead of -0).  That is the reference's order wherever its fp32 scores differ
and a defined refinement inside every group of equal fp32 scores.  The threshold `score > t` (dense_detector.py:207)
becomes its exact-arithmetic form `logit > log(t / (1 - t))`, evaluated in double for the fp32 value of t
(logit_lower_bound below).  `inference_single_image` ranks the NMS by the same logits.

Pinned against the reference's own DenseDetector methods (oracle/ref.py::py_dense_detector) through
tests/golden/dense_detector.npz (tests/test_oracle_golden.py): the selected set is the reference's, the order differs
from the reference's only inside groups of equal fp32 score, boxes / scores agree to the rounding of exp()."""
import math

import numpy as np

from . import batched_nms
from .rpn import DEFAULT_SCALE_CLAMP, apply_deltas


def sigmoid32(x):
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore"):
        return (np.float32(1) / (np.float32(1) + np.exp(-x).astype(np.float32))).astype(np.float32)


def logit_lower_bound(score_thresh):
    """Smallest fp32 logit that is a candidate: sigmoid(x) > t in exact arithmetic, t = fp32(score_thresh).
    t >= 1: none (NaN); t < 0: all; t == 0: all whose fp32 sigmoid is not flushed to zero (exp(-x) finite in fp32)."""
    t = float(np.float32(score_thresh))
    if math.isnan(t) or t >= 1.0:
        return np.float32(np.nan)
    if t < 0.0:
        return np.float32(-np.inf)
    if t == 0.0:
        return np.float32(-88.72283)
    bound = math.log(t / (1.0 - t))  # double
    f = np.float32(bound)
    if not float(f) > bound:
        f = np.nextafter(f, np.float32(np.inf), dtype=np.float32)
    return f


def select_per_level(logits, score_thresh, topk_candidates):
    """logits [A,K] -> flattened indices a*K + c of the selected candidates, best first (the RANKING RULE above)."""
    flat = np.ascontiguousarray(np.asarray(logits, np.float32).reshape(-1))
    with np.errstate(invalid="ignore"):
        cand = np.nonzero(flat >= logit_lower_bound(score_thresh))[0]
    order = np.lexsort((cand, -total_order_key(flat[cand])))  # logit desc, flattened index asc
    return cand[order][:min(len(cand), topk_candidates)]


def total_order_key(x):
    """int64 image of fp32 values in IEEE-754 total order (-inf < ... < -0 < +0 < ... < +inf): what 'logit
    descending' means bit for bit, so that +0 ranks ahead of -0 on both sides."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.int64)
    return np.where(u & 0x80000000, 0xFFFFFFFF - u, u + 0x80000000)


def decode_per_level(anchors, logits, deltas, score_thresh, topk_candidates, weights=(1.0, 1.0, 1.0, 1.0),
                     scale_clamp=DEFAULT_SCALE_CLAMP, return_logits=False):
    """anchors [A,4], logits [A,K], deltas [A,4] -> boxes [n,4], scores [n], classes [n] (best first)."""
    logits = np.asarray(logits, np.float32)
    k = logits.shape[1]
    sel = select_per_level(logits, score_thresh, topk_candidates)
    anchor_idx, cls = sel // k, sel % k
    boxes = apply_deltas(np.asarray(deltas, np.float32)[anchor_idx], np.asarray(anchors, np.float32)[anchor_idx], weights,
                         scale_clamp)
    x = logits.reshape(-1)[sel]
    out = (boxes.reshape(-1, 4), sigmoid32(x), cls.astype(np.int64))
    return out + (x,) if return_logits else out


def decode_multi_level(anchors, logits, deltas, score_thresh, topk_candidates, weights=(1.0, 1.0, 1.0, 1.0),
                       scale_clamp=DEFAULT_SCALE_CLAMP, return_logits=False):
    parts = [decode_per_level(a, l, d, score_thresh, topk_candidates, weights, scale_clamp, return_logits)
             for a, l, d in zip(anchors, logits, deltas)]
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(len(parts[0])))


def inference_single_image(anchors, logits, deltas, score_thresh, topk_candidates, nms_thresh, max_detections,
                           weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=DEFAULT_SCALE_CLAMP):
    boxes, scores, cls, x = decode_multi_level(anchors, logits, deltas, score_thresh, topk_candidates, weights,
                                               scale_clamp, return_logits=True)
    keep = batched_nms(boxes, x, cls, nms_thresh)[:max_detections]  # ranked by logit: see RANKING RULE
    return boxes[keep], scores[keep], cls[keep]
