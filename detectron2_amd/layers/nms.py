"""NMS surface -- mirrors detectron2/layers/nms.py:5-147: `nms`, `batched_nms`, `nms_rotated`,
`batched_nms_rotated`.  Contract kept: int64 indices sorted by decreasing score; inputs are not
modified.  `batched_nms*` suppress independently per category on the ORIGINAL coordinates (no
coordinate-offset trick: the device kernel is category-aware, nms.py:137-145 is not needed)."""
import torch

from .ops import nms_images, nms_impl  # noqa: F401  (importing .ops registers torch.ops.d2amd.* / detectron2.*)
from .wrappers import disable_torch_compiler


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.nms semantics (re-exported by the reference at nms.py:6): greedy NMS on
    Tensor[N,4] xyxy boxes, suppressing IoU > iou_threshold.  `torch.jit.script`-able (a registered op)."""
    return torch.ops.d2amd.nms(boxes, scores, iou_threshold)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float):
    """Same as torchvision.ops.boxes.batched_nms, but with float() (nms.py:11-22).  `torch.jit.script`-able with
    identical results (tests/layers/test_nms.py:16-29)."""
    assert boxes.shape[-1] == 4
    return torch.ops.d2amd.batched_nms(boxes.float(), scores, idxs, iou_threshold)


@disable_torch_compiler
def nms_rotated(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float):
    """Rotated NMS on Tensor[N,5] (x_ctr, y_ctr, w, h, angle_degrees) boxes (nms.py:27-89):
    iteratively removes lower scoring boxes whose IoU with a kept box is >= iou_threshold
    (the reference's CPU comparison).  Returns int64 indices in decreasing score order."""
    return torch.ops.detectron2.nms_rotated(boxes, scores, iou_threshold)


@torch.jit.script_if_tracing
def batched_nms_rotated(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float):
    """Per-category rotated NMS (nms.py:96-147; `script_if_tracing` like the reference, :96).  The device kernel is
    category-aware, so the reference's coordinate-offset trick (:137-145) is not needed."""
    assert boxes.shape[-1] == 5
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()  # fp16 does not have enough range for batched NMS
    return torch.ops.d2amd.batched_nms_rotated(boxes, scores, idxs, iou_threshold)


def batched_nms_images(inputs, iou_threshold: float, defer: bool = False, runs=None, gather=None,
                       result_buffer=None, host_mirror: bool = True):
    """`batched_nms` of every image of a batch: inputs = [(boxes [n,4], scores [n], idxs [n]), ...] ->
    list of kept-index tensors (each as `batched_nms` would return).  Replaces the per-image loop +
    per-image host sync of find_top_rpn_proposals (proposal_generator/proposal_utils.py:118-135) and
    DenseDetector._decode_multi_level_predictions / inference (meta_arch/dense_detector.py:186-260):
    the images' device pipelines overlap on separate HIP streams and there is one sync per batch.
    defer=True enqueues everything and returns a callable that performs that sync and returns the list: work that does
    not depend on the NMS (e.g. the anchor-labelling IoU of the same RPN iteration) can be enqueued in between.
    runs = (run_offsets, runs_are_categories): the rows of every image are pre-sorted runs -- the per-level top-k lists
    both callers have just built (rows parked at score -inf excepted); the order is then merged from the runs instead
    of ranked from scratch, with identical results (include/d2amd.h: d2amd_nms_runs).
    gather (with runs and defer): per image, up to 4 tensors [n, ...]; their kept rows, in keep order, are written while
    the kept indices are (no `x[keep]` launches after the sync): `.gathered` of the returned callable.
    result_buffer: an int32 device tensor of 8 * len(inputs) + k words owned by the caller, whose last k words its own
    kernels have written (status flags): results and those words reach the host in one transfer; the callable then
    always returns (kept, finite counts, the k words).  host_mirror=False: no transfer is enqueued (the caller reads the
    buffer on the device; calling the callable then costs a synchronous read)."""
    for b, _s, _i in inputs:
        assert b.shape[-1] == 4
    return nms_images([(b.float(), s, i) for b, s, i in inputs], iou_threshold, False, defer, runs, gather,
                      result_buffer, host_mirror)
