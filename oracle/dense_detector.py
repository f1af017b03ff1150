"""CPU restatement (numpy, fp32) of the dense-detector (RetinaNet) inference path -- TEST INFRASTRUCTURE ONLY.

  decode_per_level          <- detectron2/modeling/meta_arch/dense_detector.py:186-232 (_decode_per_level_predictions:
                               `scores > score_thresh`, nonzero, topk, apply_deltas of the selected anchors)
  decode_multi_level        <- dense_detector.py:234-260 (per level, concatenated)
  inference_single_image    <- detectron2/modeling/meta_arch/retinanet.py:275-309 (decode, batched_nms, top
                               max_detections), with `batched_nms` = oracle.batched_nms (the torchvision restatement)
Scores are sigmoid(logit) in fp32 as retinanet.py:267 computes them.  torch.topk leaves the order of equal scores
unspecified; this restatement (and the HIP path) break ties towards the lower flattened index a*K + c.
Pinned against the reference's own DenseDetector methods (oracle/ref.py::py_dense_detector) through
tests/golden/dense_detector.npz: selection exact, decoded boxes / scores to the rounding of exp()."""
import numpy as np

from . import batched_nms
from .rpn import DEFAULT_SCALE_CLAMP, apply_deltas


def sigmoid32(x):
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore"):
        return (np.float32(1) / (np.float32(1) + np.exp(-x).astype(np.float32))).astype(np.float32)


def decode_per_level(anchors, logits, deltas, score_thresh, topk_candidates, weights=(1.0, 1.0, 1.0, 1.0),
                     scale_clamp=DEFAULT_SCALE_CLAMP):
    """anchors [A,4], logits [A,K], deltas [A,4] -> boxes [n,4], scores [n], classes [n] (score descending)."""
    scores = sigmoid32(logits)
    a, k = scores.shape
    flat = scores.reshape(-1)
    cand = np.nonzero(flat > np.float32(score_thresh))[0]
    order = np.lexsort((cand, -flat[cand].astype(np.float64)))  # score desc, flattened index asc
    sel = cand[order][:min(len(cand), topk_candidates)]
    anchor_idx, cls = sel // k, sel % k
    boxes = apply_deltas(np.asarray(deltas, np.float32)[anchor_idx], np.asarray(anchors, np.float32)[anchor_idx], weights,
                         scale_clamp)
    return boxes.reshape(-1, 4), flat[sel], cls.astype(np.int64)


def decode_multi_level(anchors, logits, deltas, score_thresh, topk_candidates, weights=(1.0, 1.0, 1.0, 1.0),
                       scale_clamp=DEFAULT_SCALE_CLAMP):
    parts = [decode_per_level(a, l, d, score_thresh, topk_candidates, weights, scale_clamp)
             for a, l, d in zip(anchors, logits, deltas)]
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]))


def inference_single_image(anchors, logits, deltas, score_thresh, topk_candidates, nms_thresh, max_detections,
                           weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=DEFAULT_SCALE_CLAMP):
    boxes, scores, cls = decode_multi_level(anchors, logits, deltas, score_thresh, topk_candidates, weights, scale_clamp)
    keep = batched_nms(boxes, scores, cls, nms_thresh)[:max_detections]
    return boxes[keep], scores[keep], cls[keep]
