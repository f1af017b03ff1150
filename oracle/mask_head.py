"""CPU restatement (numpy) of the mask-head glue -- TEST INFRASTRUCTURE ONLY.

  mask_rcnn_loss       <- detectron2/modeling/roi_heads/mask_head.py:31-113 (gather of the gt-class plane :77-79,
                          accuracy / false positive / false negative :88-95, BCE with logits, mean :112)
  mask_rcnn_loss_grad  <- autograd of the above: (sigmoid(x) - t) * g / numel in the class plane, 0 elsewhere
  mask_rcnn_inference  <- mask_head.py:116-158 (gather of the predicted-class plane + sigmoid)
  mask_rcnn_loss_masked / _grad <- the same on the rows whose class is valid (the foreground subset of fixed-size lists)
Evaluated in float64 (the reference computes in fp32; parity bar 1e-5 relative on the loss, 1e-6 absolute on
probabilities / gradients).  Pinned against the reference's own functions run on CPU (loaded with import stubs
by oracle/ref.py::py_mask_head) through tests/golden/mask_head.npz."""
import numpy as np


def _plane(logits, classes):
    logits = np.asarray(logits, np.float64)
    b = logits.shape[0]
    if logits.shape[1] == 1:
        return logits[:, 0]
    return logits[np.arange(b), np.asarray(classes, np.int64)]


def mask_rcnn_loss(logits, gt_classes, gt_masks):
    """-> (loss, dict of the logged scalars)."""
    x = _plane(logits, gt_classes)
    t = np.asarray(gt_masks).astype(bool)
    # binary_cross_entropy_with_logits: (1 - t) * x - log_sigmoid(x);  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
    ls = np.minimum(x, 0.0) - np.log1p(np.exp(-np.abs(x)))
    loss = float(np.mean((1.0 - t) * x - ls))
    wrong = (x > 0.0) != t
    npos = int(t.sum())
    stats = {
        "accuracy": 1 - wrong.sum() / max(wrong.size, 1.0),
        "false_positive": (wrong & ~t).sum() / max(t.size - npos, 1.0),
        "false_negative": (wrong & t).sum() / max(npos, 1.0),
        "counts": np.array([wrong.sum(), npos, (wrong & ~t).sum(), (wrong & t).sum()], np.int64),
    }
    return loss, stats


def mask_rcnn_loss_grad(logits, gt_classes, gt_masks, grad_loss=1.0):
    logits = np.asarray(logits, np.float64)
    x = _plane(logits, gt_classes)
    t = np.asarray(gt_masks).astype(np.float64)
    g = (1.0 / (1.0 + np.exp(-x)) - t) * (grad_loss / x.size)
    out = np.zeros_like(logits)
    b = logits.shape[0]
    if logits.shape[1] == 1:
        out[:, 0] = g
    else:
        out[np.arange(b), np.asarray(gt_classes, np.int64)] = g
    return out


def mask_rcnn_loss_masked(logits, gt_classes, gt_masks):
    """The loss over the rows whose class lies in [0, C): what mask_rcnn_loss (mask_head.py:47-113) computes on the
    foreground subset select_foreground_proposals (roi_heads.py:37-75) hands it, for fixed-size lists that carry
    background (class C) and padding (-1) rows.  -> (loss, stats, rows that count); no row -> 0 (mask_head.py:71-72)."""
    logits = np.asarray(logits, np.float64)
    cls = np.asarray(gt_classes, np.int64)
    ok = (cls >= 0) & (cls < logits.shape[1])
    if not ok.any():
        return 0.0, {"counts": np.zeros(4, np.int64)}, 0
    loss, stats = mask_rcnn_loss(logits[ok], cls[ok], np.asarray(gt_masks)[ok])
    return loss, stats, int(ok.sum())


def mask_rcnn_loss_masked_grad(logits, gt_classes, gt_masks, grad_loss=1.0):
    logits = np.asarray(logits, np.float64)
    cls = np.asarray(gt_classes, np.int64)
    ok = (cls >= 0) & (cls < logits.shape[1])
    out = np.zeros_like(logits)
    if ok.any():
        out[ok] = mask_rcnn_loss_grad(logits[ok], cls[ok], np.asarray(gt_masks)[ok], grad_loss)
    return out


def mask_rcnn_inference(logits, classes):
    x = _plane(logits, classes)
    return (1.0 / (1.0 + np.exp(-x)))[:, None]
