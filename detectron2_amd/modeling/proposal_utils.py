"""RPN / RetinaNet proposal selection -- the caller of `batched_nms`.

`find_top_rpn_proposals_fused` = `RPN._decode_proposals` (proposal_generator/rpn.py:468-533) followed by
`find_top_rpn_proposals` (proposal_generator/proposal_utils.py:22-135) for the whole batch:
  per (image, level) top-k of the objectness logits -> decode only the selected anchors
  (Box2BoxTransform.apply_deltas, modeling/box_regression.py:71-116) -> clip -> finite / non-empty
  filter (d2amd_rpn_select_proposals, one call) -> per-image, per-level NMS of all images on overlapping
  HIP streams (batched_nms_images) -> the post_nms_topk best per image.
The reference loops over levels and images in Python with a device->host sync per image (`.item()`,
NMS result size); here there is one sync per batch (NMS counts + valid counts + non-finite flag in one transfer).
Results are equal to the reference's up to the rounding of exp() in the decode (the selection and NMS
run on identical inputs otherwise); ties between equal logits resolve towards the lower anchor index."""
import ctypes
import math
from typing import List, Tuple

import torch

from .. import _C
from ..layers.nms import batched_nms_images
from ..structures import Boxes

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Proposals:
    """Minimal stand-in for `Instances` (out of scope): the two fields find_top_rpn_proposals sets."""

    def __init__(self, image_size, proposal_boxes, objectness_logits):
        self.image_size = image_size
        self.proposal_boxes = proposal_boxes
        self.objectness_logits = objectness_logits

    def __len__(self):
        return len(self.proposal_boxes)


def rpn_select_proposals(anchors: List[torch.Tensor], pred_objectness_logits: List[torch.Tensor],
                         pred_anchor_deltas: List[torch.Tensor], image_sizes: List[Tuple[int, int]],
                         pre_nms_topk: int, min_box_size: float, weights=(1.0, 1.0, 1.0, 1.0),
                         scale_clamp: float = _DEFAULT_SCALE_CLAMP):
    """Steps 1-2 of find_top_rpn_proposals + decode + clip + validity for all images.
    anchors[l] [A_l,4]; pred_objectness_logits[l] [N,A_l]; pred_anchor_deltas[l] [N,A_l,4].
    Returns boxes [N,K,4], scores [N,K], valid [N,K] bool, level_ids [K] int64, flags [1] int32 (device)."""
    _C.require_gpu(*anchors, *pred_objectness_logits, *pred_anchor_deltas, op="rpn_select_proposals")
    sizes = [int(a.shape[0]) for a in anchors]
    n = int(pred_objectness_logits[0].shape[0])
    dev = anchors[0].device
    logits = torch.cat([t.detach().float() for t in pred_objectness_logits], dim=1).contiguous()
    deltas = torch.cat([t.detach().float() for t in pred_anchor_deltas], dim=1).contiguous()
    anc = torch.cat([a.detach().float() for a in anchors], dim=0).contiguous()
    atot = sum(sizes)
    assert logits.shape == (n, atot) and deltas.shape == (n, atot, 4) and len(image_sizes) == n
    k = sum(min(s, pre_nms_topk) for s in sizes)
    boxes = torch.empty((n, k, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((n, k), dtype=torch.float32, device=dev)
    valid = torch.empty((n, k), dtype=torch.uint8, device=dev)
    level_ids = torch.empty((k,), dtype=torch.int64, device=dev)
    flags = torch.zeros((1,), dtype=torch.int32, device=dev)
    if n == 0 or k == 0:
        return boxes, scores, valid.bool(), level_ids, flags
    L = _C.lib()
    lv = (ctypes.c_int * len(sizes))(*sizes)
    hw = (ctypes.c_int * (2 * n))(*[int(v) for s in image_sizes for v in s])
    wts = (ctypes.c_float * 4)(*[float(v) for v in weights])
    with _C.on_device(dev):
        ws_bytes = L.d2amd_rpn_select_workspace_bytes(n, atot)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _C.check(L.d2amd_rpn_select_proposals(_C.ptr(logits), _C.ptr(deltas), _C.ptr(anc), n, atot, lv, len(sizes), hw,
                                              int(pre_nms_topk), float(min_box_size), wts, float(scale_clamp),
                                              _C.ptr(boxes), _C.ptr(scores), _C.ptr(valid), _C.ptr(level_ids),
                                              _C.ptr(flags), _C.ptr(ws), ws_bytes, _C.stream()))
    return boxes, scores, valid.bool(), level_ids, flags


def find_top_rpn_proposals_fused(anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes, nms_thresh: float,
                                 pre_nms_topk: int, post_nms_topk: int, min_box_size: float, training: bool,
                                 weights=(1.0, 1.0, 1.0, 1.0), scale_clamp: float = _DEFAULT_SCALE_CLAMP,
                                 defer: bool = False):
    """-> list of N `Proposals` (proposal_boxes: Boxes, objectness_logits), sorted by objectness.
    ONE host sync per batch: the kept counts, the number of kept boxes that are valid proposals (invalid rows are
    parked at score -inf and sort last: the NMS reports how many kept boxes have a finite score) and the non-finite
    flag come back in a single transfer.  defer=True enqueues everything and returns a callable that performs that
    sync and builds the list: work that does not depend on the proposals (the anchor labelling of the same RPN
    iteration, rpn.py:431-480) can be enqueued in between."""
    boxes, scores, valid, level_ids, flags = rpn_select_proposals(
        anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes, pre_nms_topk, min_box_size, weights,
        scale_clamp)
    n = boxes.shape[0]
    # invalid rows are zero-area boxes with score -inf: they neither suppress nor get suppressed and sort last
    nms_done = batched_nms_images([(boxes[i], scores[i], level_ids) for i in range(n)], nms_thresh, defer=True)

    def finish():
        keeps, n_finite, (bad,) = nms_done(with_finite=True, extra=flags) if n else ([], [], (0,))  # the one sync
        if n and bad and training:
            raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
        out = []
        for i, k in enumerate(keeps):
            k = k[:min(post_nms_topk, n_finite[i])]
            out.append(Proposals(tuple(image_sizes[i]), Boxes(boxes[i][k]), scores[i][k]))
        return out

    return finish if defer else finish()
