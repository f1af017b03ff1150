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


class DeviceProposals:
    """The proposals of a batch with their counts left on the device (no host sync): per image `boxes`
    [post_nms_topk, 4] / `logits` [post_nms_topk] in objectness order, valid up to min(post_nms_topk, limits[0],
    limits[2]) -- `limits[i]` is the NMS result row {kept, flags, finite, 0} (int64 words, limit_stride 2) -- and
    `nonfinite_flag` (int32[1]; non-zero: predicted boxes or scores contained Inf / NaN, what the synchronous path
    raises FloatingPointError for: check it after the step).  limits[i][1] != 0 (NMS flags: the synchronous path redoes
    or rejects that image) comes with limits[i][2] = 0: such an image contributes no proposals on the device path.
    `find_top_rpn_proposals_fused(...).device` is None when the NMS did not run as the batched pipeline (more than
    12,288 boxes in an image, mixed devices): its counts are then not in this buffer."""

    def __init__(self, boxes, logits, limits, nonfinite_flag, image_sizes):
        self.boxes, self.logits, self.limits, self.nonfinite_flag = boxes, logits, limits, nonfinite_flag
        self.image_sizes = image_sizes

    def counts(self):
        """the valid lengths as a device tensor [N] (int64; torch ops, no sync)"""
        cap = self.boxes[0].shape[0] if self.boxes else 0
        lim = torch.stack(self.limits)
        return torch.minimum(lim[:, 0], lim[:, 2]).clamp(min=0, max=cap)

    def pad_(self):
        """rows at / behind the live count <- the box (0, 0, 1, 1), in place and without a host read (one launch): what the
        box pooler and `fast_rcnn_predict(limits=self.limits)` take at the fixed shape [post_nms_topk, 4].  -> self.boxes"""
        n = len(self.boxes)
        if n == 0 or self.boxes[0].shape[0] == 0:
            return self.boxes
        assert all(b.is_contiguous() and b.dtype == torch.float32 for b in self.boxes)
        assert all(l.is_contiguous() and l.dtype == torch.int64 for l in self.limits)
        vp = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        with _C.on_device(self.boxes[0].device):
            _C.check(_C.lib().d2amd_proposals_pad(vp(self.boxes), vp(self.limits), (ctypes.c_int * n)(*[int(b.shape[0]) for b in self.boxes]),
                                                  n, _C.stream()))
        return self.boxes


def rpn_select_proposals(anchors: List[torch.Tensor], pred_objectness_logits: List[torch.Tensor],
                         pred_anchor_deltas: List[torch.Tensor], image_sizes: List[Tuple[int, int]],
                         pre_nms_topk: int, min_box_size: float, weights=(1.0, 1.0, 1.0, 1.0),
                         scale_clamp: float = _DEFAULT_SCALE_CLAMP, flags_out: torch.Tensor = None):
    """Steps 1-2 of find_top_rpn_proposals + decode + clip + validity for all images.
    anchors[l] [A_l,4]; pred_objectness_logits[l] [N,A_l]; pred_anchor_deltas[l] [N,A_l,4].
    Returns boxes [N,K,4], scores [N,K], valid [N,K] bool, level_ids [K] int64, flags [1] int32 (device)."""
    _C.require_gpu(*anchors, *pred_objectness_logits, *pred_anchor_deltas, op="rpn_select_proposals")
    sizes = [int(a.shape[0]) for a in anchors]
    n = int(pred_objectness_logits[0].shape[0])
    dev = anchors[0].device
    # the head's per-level tensors go to the device as they are (no torch.cat: 2 x 268,569 x 9 floats per call)
    logits = [t.detach().float().contiguous() for t in pred_objectness_logits]
    deltas = [t.detach().float().contiguous() for t in pred_anchor_deltas]
    anc = [a.detach().float().contiguous() for a in anchors]
    atot = sum(sizes)
    for l, s_l in enumerate(sizes):
        assert logits[l].shape == (n, s_l) and deltas[l].shape == (n, s_l, 4) and anc[l].shape == (s_l, 4), \
            (l, logits[l].shape, deltas[l].shape, anc[l].shape)
    assert len(image_sizes) == n
    k = sum(min(s, pre_nms_topk) for s in sizes)
    boxes = torch.empty((n, k, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((n, k), dtype=torch.float32, device=dev)
    valid = torch.empty((n, k), dtype=torch.bool, device=dev)  # written as 0 / 1 bytes
    level_ids = torch.empty((k,), dtype=torch.int64, device=dev)
    # zeroed by the call (flags_out: the caller's int32[1], e.g. a word of its result buffer)
    flags = flags_out if flags_out is not None else torch.empty((1,), dtype=torch.int32, device=dev)
    if n == 0 or k == 0:
        return boxes, scores, valid, level_ids, flags.zero_()
    L = _C.lib()
    nl = len(sizes)
    lv = (ctypes.c_int * nl)(*sizes)
    hw = (ctypes.c_int * (2 * n))(*[int(v) for s in image_sizes for v in s])
    wts = (ctypes.c_float * 4)(*[float(v) for v in weights])
    parr = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])
    with _C.on_device(dev):
        ws_bytes = L.d2amd_rpn_select_workspace_bytes(n, atot)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        # (pre_nms_topk > 65,536 per level: D2AMD_EUNSUPPORTED -> RuntimeError; the reference's configurations use
        # 1,000 .. 12,000)
        _C.check(L.d2amd_rpn_select_proposals_levels(parr(logits), parr(deltas), parr(anc), n, lv, nl, hw,
                                                     int(pre_nms_topk), float(min_box_size), wts,
                                                     float(scale_clamp), _C.ptr(boxes), _C.ptr(scores),
                                                     _C.ptr(valid), _C.ptr(level_ids), _C.ptr(flags), _C.ptr(ws),
                                                     ws_bytes, _C.stream()))
    return boxes, scores, valid, level_ids, flags


def find_top_rpn_proposals_fused(anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes, nms_thresh: float,
                                 pre_nms_topk: int, post_nms_topk: int, min_box_size: float, training: bool,
                                 weights=(1.0, 1.0, 1.0, 1.0), scale_clamp: float = _DEFAULT_SCALE_CLAMP,
                                 defer: bool = False, beside_nms=None, join_beside: bool = True,
                                 host_result: bool = True):
    """-> list of N `Proposals` (proposal_boxes: Boxes, objectness_logits), sorted by objectness.
    ONE host sync per batch: the kept counts, the number of kept boxes that are valid proposals (invalid rows are
    parked at score -inf and sort last: the NMS reports how many kept boxes have a finite score) and the non-finite
    flag come back in a single transfer.  defer=True enqueues everything and returns a callable that performs that
    sync and builds the list: work that does not depend on the proposals (the anchor labelling of the same RPN
    iteration, rpn.py:431-480) can be enqueued in between.
    beside_nms: a callable enqueuing such independent work; it runs on a side stream BESIDE THE NMS (forked after the
    selection + decode, joined before this function returns; its result is `.beside` of the returned callable /
    ignored without defer).  The selection kernels are latency-bound chains of few workgroups that slow down when
    chip-filling kernels run next to them (fused select 27 -> 35 us, decode 5 -> 13 us beside the anchor matcher); the
    NMS reduction (10 workgroups walking their lists) does not.  Captured RPN half of bench.py: 159.5 -> 156.5 us.
    host_result=False: no device-to-host transfer of the counts is enqueued (a step that reads them on the device:
    `.device` of the returned callable).
    join_beside=False (with defer): the side branch is NOT joined here -- `.join_beside()` of the returned callable does
    it, whenever its results are needed (bench.py's connected step: at the end of the forward; the anchor labelling +
    sampling take longer than the NMS, and nothing before the losses reads them)."""
    n = int(pred_objectness_logits[0].shape[0])
    # one int32 buffer for everything the host reads back: 8 words per image of NMS results + the non-finite flag
    res = torch.empty(8 * n + 1, dtype=torch.int32, device=anchors[0].device)
    boxes, scores, valid, level_ids, flags = rpn_select_proposals(
        anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes, pre_nms_topk, min_box_size, weights,
        scale_clamp, flags_out=res[8 * n:])
    # invalid rows are zero-area boxes with score -inf: they neither suppress nor get suppressed and sort last.
    # The rows of an image are the per-level top-k lists, each already in score order, and the NMS is per level:
    # the order is merged from those runs (runs = categories) instead of ranked from scratch
    run_offsets = [0]
    for a in anchors:
        run_offsets.append(run_offsets[-1] + min(int(a.shape[0]), pre_nms_topk))
    def nms():
        return batched_nms_images([(boxes[i], scores[i], None) for i in range(n)], nms_thresh, defer=True,
                                  runs=(run_offsets, True), gather=[(boxes[i], scores[i]) for i in range(n)],
                                  result_buffer=res, host_mirror=host_result)

    beside = None
    if beside_nms is not None:
        from ..streams import fork_join

        nms_done, beside, *late = fork_join(nms, beside_nms, device=res.device, current_first=True,
                                            defer_join=not join_beside)
    else:
        nms_done, late = nms(), []

    def finish():
        keeps, n_finite, (bad,) = nms_done(with_finite=True) if n else ([], [], (0,))  # the one sync
        if n and bad and training:
            raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
        out = []
        for i, k in enumerate(keeps):
            m = min(post_nms_topk, n_finite[i], len(k))  # the kept rows arrive in keep order: views, no index launch
            kb, ks = nms_done.gathered[i]
            out.append(Proposals(tuple(image_sizes[i]), Boxes(kb[:m]), ks[:m]))
        return out

    finish.beside = beside
    finish.join_beside = late[0] if late else (lambda: None)
    # what the ROI heads of a captured step read instead of calling finish(): fixed-size lists whose valid length the
    # DEVICE knows (label_and_sample_proposals_fixed(limits=..., limit_stride=2))
    # Only the batched NMS pipeline fills `res` (<= RANK_MAX_N = 12,288 boxes per image, one device): the per-image
    # fallback keeps its results in its own buffers, so there is nothing a device-side consumer could read -> None.
    # A raised NMS flag (limits[i][1] != 0: the host path redoes or rejects that image) zeroes the image's `finite`
    # word on the device, i.e. min(kept, finite) = 0 proposals -- never rows of a wrong order.
    finish.device = None
    if n == 0 or getattr(nms_done, "result_on_device", False):
        finish.device = DeviceProposals(
            [nms_done.gathered[i][0][:post_nms_topk] for i in range(n)] if n else [],
            [nms_done.gathered[i][1][:post_nms_topk] for i in range(n)] if n else [],
            list(res[:8 * n].view(torch.int64).view(n, 4)) if n else [], res[8 * n:], [tuple(s) for s in image_sizes])
    return finish if defer else finish()
