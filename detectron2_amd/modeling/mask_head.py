"""Mask-head glue of `detectron2.modeling.roi_heads.mask_head` on the hot path:
`mask_rcnn_loss` (mask_head.py:31-113) and `mask_rcnn_inference` (mask_head.py:116-158), same names, arguments
and results, each backed by fused HIP kernels (csrc/mask_head.hip):

* the class plane of every ROI is read in place -- no `pred_mask_logits[indices, gt_classes]` copy, no fp32 copy
  of the targets;
* loss + the three logged statistics come out of one pass and are read with ONE host transfer (the reference
  issues four `.item()` syncs), and only when an event storage is there to receive them;
* the backward writes the full (B, C, M, M) gradient once (zeros outside the class planes).
"""
import ctypes
from typing import List

import torch

from .. import _C

__all__ = ["mask_rcnn_loss", "mask_rcnn_inference", "mask_rcnn_loss_from_targets"]


class _MaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gt_classes, gt_masks):
        b, c, h, w = logits.shape
        x = logits.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        stats = torch.empty(5, dtype=torch.int64, device=x.device)
        L = _C.lib()
        ws_bytes = L.d2amd_mask_rcnn_loss_workspace_bytes(b)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with _C.on_device(x.device):
            _C.check(L.d2amd_mask_rcnn_loss_forward(_C.ptr(x), _C.ptr(gt_classes), _C.ptr(gt_masks), b, c, h * w,
                                                    _C.dtype_code(x), _C.ptr(loss), _C.ptr(stats), _C.ptr(ws),
                                                    ctypes.c_size_t(ws_bytes), _C.stream()))
        ctx.save_for_backward(x, gt_classes, gt_masks)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)  # (no zero-fill launch for the statistics' "gradient")
        return loss, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss, _grad_stats):
        if grad_loss is None:
            return None, None, None
        x, gt_classes, gt_masks = ctx.saved_tensors
        b, c, h, w = x.shape
        g = grad_loss.detach().to(dtype=torch.float32).contiguous()
        grad = torch.empty_like(x)
        with _C.on_device(x.device):
            _C.check(_C.lib().d2amd_mask_rcnn_loss_backward(_C.ptr(x), _C.ptr(gt_classes), _C.ptr(gt_masks), _C.ptr(g),
                                                            b, c, h * w, _C.dtype_code(x), _C.ptr(grad), _C.stream()))
        return grad, None, None


class _MaskLossMasked(torch.autograd.Function):
    """_MaskLoss over the rows whose class lies in [0, C) (d2amd_mask_rcnn_loss_forward_masked / _backward_masked):
    the background / padding rows of the fixed-size lists of `label_and_sample_proposals_fixed` do not count."""

    @staticmethod
    def forward(ctx, logits, gt_classes, gt_masks):
        b, c, h, w = logits.shape
        x = logits.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        stats = torch.empty(6, dtype=torch.int64, device=x.device)
        L = _C.lib()
        ws_bytes = L.d2amd_mask_rcnn_loss_workspace_bytes(b)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with _C.on_device(x.device):
            _C.check(L.d2amd_mask_rcnn_loss_forward_masked(_C.ptr(x), _C.ptr(gt_classes), _C.ptr(gt_masks), b, c, h * w,
                                                           _C.dtype_code(x), _C.ptr(loss), _C.ptr(stats), _C.ptr(ws),
                                                           ctypes.c_size_t(ws_bytes), _C.stream()))
        # the row count stays on the device; the backward reads it out of `stats` (saved: autograd's version check
        # refuses a backward after an in-place change -- a copy of the one word was a launch at the end of the forward)
        ctx.save_for_backward(x, gt_classes, gt_masks, stats)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return loss, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss, _grad_stats):
        if grad_loss is None:
            return None, None, None
        x, gt_classes, gt_masks, stats = ctx.saved_tensors
        b, c, h, w = x.shape
        g = grad_loss.detach().to(dtype=torch.float32).contiguous()
        grad = torch.empty_like(x)
        with _C.on_device(x.device):
            _C.check(_C.lib().d2amd_mask_rcnn_loss_backward_masked(
                _C.ptr(x), _C.ptr(gt_classes), _C.ptr(gt_masks), _C.ptr(g), ctypes.c_void_p(stats.data_ptr() + 40), b, c,
                h * w,
                _C.dtype_code(x), _C.ptr(grad), _C.stream()))
        return grad, None, None


def mask_rcnn_loss_from_targets(pred_mask_logits: torch.Tensor, gt_classes, gt_masks: torch.Tensor,
                                ignore_invalid_rows: bool = False):
    """Fused core of `mask_rcnn_loss`: (B, C, M, M) logits, (B,) int64 gt classes (ignored / may be None when
    C == 1), (B, M, M) bool targets -> (loss fp32 scalar, stats int64[5] on the device:
    #incorrect, #positive, #false positive, #false negative, #rows with a class outside [0, C)).
    ignore_invalid_rows: rows with a class outside [0, C) -- background and
    padding rows of `label_and_sample_proposals_fixed` -- do not count (mean over the others, zero gradient);
    stats int64[6]: [4] = ignored rows, [5] = rows that count."""
    _C.require_gpu(pred_mask_logits, gt_masks, op="mask_rcnn_loss")
    b, c = pred_mask_logits.shape[:2]
    assert pred_mask_logits.size(2) == pred_mask_logits.size(3), "Mask prediction must be square!"
    assert gt_masks.shape == (b,) + tuple(pred_mask_logits.shape[2:]), (gt_masks.shape, pred_mask_logits.shape)
    if gt_masks.dtype == torch.bool:
        t = gt_masks.contiguous().view(torch.uint8)
    elif gt_masks.dtype == torch.uint8:
        t = gt_masks.contiguous()
    else:  # float targets are allowed by the reference (mask_head.py:81-86)
        t = (gt_masks > 0.5).contiguous().view(torch.uint8)
    t = t.to(pred_mask_logits.device)
    cls = None
    if c != 1:
        cls = gt_classes.to(device=pred_mask_logits.device, dtype=torch.int64).contiguous()
        assert cls.shape == (b,), cls.shape
    if ignore_invalid_rows:
        assert cls is not None, "ignore_invalid_rows: the classes carry the mask (class-specific logits)"
        return _MaskLossMasked.apply(pred_mask_logits, cls, t)
    return _MaskLoss.apply(pred_mask_logits, cls, t)


def _event_storage():
    try:
        from detectron2.utils.events import get_event_storage  # the reference's logger, when it is installed
    except ImportError:
        return None
    try:
        return get_event_storage()
    except AssertionError:
        return None


def mask_rcnn_loss(pred_mask_logits: torch.Tensor, instances: List, vis_period: int = 0, storage=None):
    """Mask R-CNN mask loss; `instances` are the reference's `Instances` (fields gt_classes, gt_masks with
    `crop_and_resize`, proposal_boxes) or any objects with those attributes.  Logs mask_rcnn/accuracy,
    false_positive, false_negative to `storage` (default: detectron2's current EventStorage if importable)."""
    cls_agnostic_mask = pred_mask_logits.size(1) == 1
    mask_side_len = pred_mask_logits.size(2)
    assert pred_mask_logits.size(2) == pred_mask_logits.size(3), "Mask prediction must be square!"
    from ..structures.masks import BitMasks, crop_and_resize_batch

    gt_classes, per_img = [], []
    for per_image in instances:
        if len(per_image) == 0:
            continue
        if not cls_agnostic_mask:
            gt_classes.append(per_image.gt_classes.to(dtype=torch.int64))
        per_img.append((per_image.gt_masks, per_image.proposal_boxes.tensor))
    if len(per_img) == 0:
        return pred_mask_logits.sum() * 0
    dev = pred_mask_logits.device
    if (len(per_img) <= 64 and all(isinstance(m, BitMasks) and m.tensor.device == dev for m, _ in per_img)
            and len({tuple(m.image_size) for m, _ in per_img}) == 1):
        # bitmask targets of the whole batch in one launch (the reference loops over the images and concatenates)
        gt_masks = crop_and_resize_batch([m for m, _ in per_img], [b for _, b in per_img], mask_side_len)
    else:
        gt_masks = torch.cat([m.crop_and_resize(b, mask_side_len).to(device=dev) for m, b in per_img], dim=0)
    cls = None if cls_agnostic_mask else torch.cat(gt_classes, dim=0)
    loss, stats = mask_rcnn_loss_from_targets(pred_mask_logits, cls, gt_masks)
    storage = storage if storage is not None else _event_storage()
    if storage is None:
        # nobody reads the statistics (no host sync here): a gt class outside [0, C) -- an IndexError in the
        # reference's gather (mask_head.py:78-79) -- must still not pass silently: it turns the loss into NaN
        loss = torch.where(stats[4] > 0, torch.full_like(loss, float("nan")), loss)
    if storage is not None:
        incorrect, positive, false_pos, false_neg, bad = stats.tolist()  # the one host sync
        if bad:
            raise IndexError(f"mask_rcnn_loss: {bad} gt_classes outside [0, {pred_mask_logits.size(1)})")
        numel = gt_masks.numel()
        storage.put_scalar("mask_rcnn/accuracy", 1 - incorrect / max(numel, 1.0))
        storage.put_scalar("mask_rcnn/false_positive", false_pos / max(numel - positive, 1.0))
        storage.put_scalar("mask_rcnn/false_negative", false_neg / max(positive, 1.0))
    return loss


def mask_rcnn_inference(pred_mask_logits: torch.Tensor, pred_instances: List):
    """Attach `pred_masks` (n_i, 1, M, M) = sigmoid of the predicted-class plane to every element of
    `pred_instances` (objects with `pred_classes` and `__len__`)."""
    _C.require_gpu(pred_mask_logits, op="mask_rcnn_inference")
    b, c, h, w = pred_mask_logits.shape
    x = pred_mask_logits.detach().contiguous()
    cls = None
    if c != 1:
        cls = torch.cat([i.pred_classes for i in pred_instances]).to(device=x.device, dtype=torch.int64).contiguous()
        assert cls.shape == (b,), (cls.shape, b)
    out = torch.empty((b, 1, h, w), dtype=x.dtype, device=x.device)
    if b:
        with _C.on_device(x.device):
            _C.check(_C.lib().d2amd_mask_rcnn_inference(_C.ptr(x), _C.ptr(cls), b, c, h * w, _C.dtype_code(x),
                                                        _C.ptr(out), _C.stream()))
    num_boxes_per_image = [len(i) for i in pred_instances]
    for prob, inst in zip(out.split(num_boxes_per_image, dim=0), pred_instances):
        inst.pred_masks = prob  # (n_i, 1, Hmask, Wmask)
