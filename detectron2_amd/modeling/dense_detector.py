"""Dense-detector (RetinaNet) inference in front of and including NMS (second half) -- the fused
counterpart of `DenseDetector._decode_multi_level_predictions` (meta_arch/dense_detector.py:186-260) and
`RetinaNet.inference_single_image` (meta_arch/retinanet.py:257-309) for the WHOLE batch:

  per (image, level): candidates = (anchor, class) pairs with sigmoid(logit) > score_thresh, the
  min(topk_candidates, #candidates) best by a segmented radix select, decode of the selected anchors
  (d2amd_dense_select_predictions: one call, no host sync -- the reference runs `sigmoid_` over all N x 16 M class
  logits, then per level and image `nonzero` [a sync] + `topk` + gathers) -> per-image, per-class NMS of all images in
  one call (batched_nms_images) -> the max_detections best per image.
Ranking is by LOGIT (sigmoid is monotone): the reference's `topk` order wherever its fp32 scores differ, and a defined
order -- higher logit, then lower flattened (anchor, class) index -- inside a group of equal fp32 scores, where
torch.topk's is unspecified.  It does not depend on any exp() implementation; the reported scores are sigmoid(logit)
of the selected rows and equal the reference's up to the rounding of exp().  The threshold `score > t`
(dense_detector.py:207) is applied in its exact form `logit > log(t / (1 - t))` (double, fp32 t)."""
import ctypes
import math
from typing import List

import torch

from .. import _C
from ..layers.nms import batched_nms_images
from ..structures import Boxes

__all__ = ["Detections", "dense_select_predictions", "dense_detector_inference_fused"]

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Detections:
    """Minimal stand-in for `Instances` (out of scope): the fields the dense-detector inference sets."""

    def __init__(self, image_size, pred_boxes, scores, pred_classes):
        self.image_size = image_size
        self.pred_boxes = pred_boxes
        self.scores = scores
        self.pred_classes = pred_classes

    def __len__(self):
        return len(self.pred_boxes)


def _ptrs(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def dense_select_predictions(anchors: List[torch.Tensor], pred_logits: List[torch.Tensor],
                             pred_anchor_deltas: List[torch.Tensor], score_thresh: float, topk_candidates: int,
                             weights=(1.0, 1.0, 1.0, 1.0), scale_clamp: float = _DEFAULT_SCALE_CLAMP,
                             return_logits: bool = False):
    """anchors[l] [A_l,4]; pred_logits[l] [N,A_l,K] class LOGITS (not probabilities); pred_anchor_deltas[l] [N,A_l,4].
    Returns boxes [N,Ktot,4], scores [N,Ktot], classes [N,Ktot] int64, valid [N,Ktot] bool, counts [N,L] int32, all on
    the device; rows of a level are best first, rows past its count are zero boxes with score -inf.  With
    return_logits, a sixth tensor [N,Ktot]: the selected logits (the ranking key; -inf past the count)."""
    _C.require_gpu(*anchors, *pred_logits, *pred_anchor_deltas, op="dense_select_predictions")
    nl = len(anchors)
    assert nl == len(pred_logits) == len(pred_anchor_deltas) and nl >= 1
    n, k_cls = int(pred_logits[0].shape[0]), int(pred_logits[0].shape[2])
    dev = anchors[0].device
    lg = [t.detach().float().contiguous() for t in pred_logits]
    dl = [t.detach().float().contiguous() for t in pred_anchor_deltas]
    an = [a.detach().float().contiguous() for a in anchors]
    sizes = [int(a.shape[0]) for a in an]
    for l in range(nl):
        assert lg[l].shape == (n, sizes[l], k_cls) and dl[l].shape == (n, sizes[l], 4), (lg[l].shape, dl[l].shape)
    ktot = sum(min(s * k_cls, int(topk_candidates)) for s in sizes)
    boxes = torch.empty((n, ktot, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((n, ktot), dtype=torch.float32, device=dev)
    classes = torch.empty((n, ktot), dtype=torch.int64, device=dev)
    valid = torch.empty((n, ktot), dtype=torch.bool, device=dev)  # written as 0 / 1 bytes
    counts = torch.zeros((n, nl), dtype=torch.int32, device=dev)
    sel_logits = torch.empty((n, ktot), dtype=torch.float32, device=dev) if return_logits else None
    if n == 0 or ktot == 0:
        out = (boxes, scores, classes, valid, counts)
        return out + (sel_logits,) if return_logits else out
    L = _C.lib()
    lv = (ctypes.c_int * nl)(*sizes)
    wts = (ctypes.c_float * 4)(*[float(v) for v in weights])
    with _C.on_device(dev):
        ws_bytes = L.d2amd_dense_select_workspace_bytes(n, lv, nl, k_cls, int(topk_candidates))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _C.check(L.d2amd_dense_select_predictions(_ptrs(lg), _ptrs(dl), _ptrs(an), n, lv, nl, k_cls, float(score_thresh),
                                                  int(topk_candidates), wts, float(scale_clamp), _C.ptr(boxes),
                                                  _C.ptr(scores), _C.ptr(classes), _C.ptr(valid), _C.ptr(counts),
                                                  _C.ptr(sel_logits) if return_logits else None, _C.ptr(ws),
                                                  ws_bytes, _C.stream()))
    out = (boxes, scores, classes, valid, counts)
    return out + (sel_logits,) if return_logits else out


def dense_detector_inference_fused(anchors, pred_logits, pred_anchor_deltas, image_sizes, score_thresh: float,
                                   topk_candidates: int, nms_thresh: float, max_detections: int,
                                   weights=(1.0, 1.0, 1.0, 1.0), scale_clamp: float = _DEFAULT_SCALE_CLAMP,
                                   defer: bool = False):
    """-> list of N `Detections` (pred_boxes: Boxes, scores, pred_classes), score-descending, at most
    max_detections each (retinanet.py:297-309).  One host sync per batch.
    defer=True: everything is ENQUEUED (no host sync: the call can be captured in a HIP graph) and a callable is
    returned that performs the one sync and builds the list."""
    boxes, scores, classes, valid, _, rank = dense_select_predictions(
        anchors, pred_logits, pred_anchor_deltas, score_thresh, topk_candidates, weights, scale_clamp, return_logits=True)
    n = boxes.shape[0]
    # rows past a level's count are zero-area boxes with score -inf: they neither suppress nor get suppressed, sort last.
    # The NMS ranks by the selected LOGITS (same order as the scores, but independent of the exp() rounding)
    # (rows past a level's count carry logit -inf: the NMS reports how many kept boxes have a finite ranking score,
    # and those sort last -- no second transfer for the valid counts)
    # (the rows are the per-level top-k lists, each in logit order: large inputs merge the order from those runs)
    run_offsets = [0]
    for a, lg in zip(anchors, pred_logits):
        run_offsets.append(run_offsets[-1] + min(int(a.shape[0]) * int(lg.shape[-1]), topk_candidates))
    if n == 0:
        return (lambda: []) if defer else []
    nms_done = batched_nms_images([(boxes[i], rank[i], classes[i]) for i in range(n)], nms_thresh, defer=True,
                                  runs=(run_offsets, False, int(pred_logits[0].shape[-1])),
                                  gather=[(boxes[i], scores[i], classes[i]) for i in range(n)])

    def finish():
        keeps, n_finite, _ = nms_done(with_finite=True)  # the one sync
        out = []
        for i, k in enumerate(keeps):
            m = min(max_detections, n_finite[i], len(k))  # the kept rows arrive in keep order: views, no index launch
            kb, ks, kc = nms_done.gathered[i]
            out.append(Detections(tuple(image_sizes[i]), Boxes(kb[:m]), ks[:m], kc[:m]))
        return out

    return finish if defer else finish()
