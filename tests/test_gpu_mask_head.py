"""GPU parity of the mask-head glue (detectron2_amd/csrc/mask_head.hip) through the C ABI:
mask_rcnn_loss forward / backward and mask_rcnn_inference against (a) the reference's own functions run on CPU
(tests/golden/mask_head.npz) and (b) the float64 restatement oracle/mask_head.py on seeded inputs.
Bars: loss 1e-5 relative; fp32 probabilities / gradients 1e-6 absolute (of a unit-scale quantity); counts exact;
16-bit I/O: one rounding of the I/O dtype (bf16 2^-8, f16 2^-11 relative) on top."""
import os

import numpy as np
import pytest
import torch

from detectron2_amd.modeling import mask_rcnn_inference, mask_rcnn_loss, mask_rcnn_loss_from_targets
from oracle import mask_head as omh

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Inst:
    """Duck-typed stand-in for detectron2.structures.Instances."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __len__(self):
        for v in self.__dict__.values():
            return len(v)
        return 0


class Targets:
    def __init__(self, t):
        self.t = t

    def __len__(self):
        return len(self.t)

    def crop_and_resize(self, boxes, side):
        assert side == self.t.shape[-1] and len(boxes) == len(self.t)
        return self.t


class Boxes:
    def __init__(self, n):
        self.tensor = torch.zeros(n, 4, device=DEV)

    def __len__(self):
        return len(self.tensor)


class Recorder:
    def __init__(self):
        self.scalars = {}

    def put_scalar(self, k, v):
        self.scalars[k] = v


def _instances(cls, gt, per_img):
    out, o = [], 0
    for n in per_img:
        out.append(Inst(gt_classes=torch.from_numpy(cls[o:o + n]).to(DEV), gt_masks=Targets(torch.from_numpy(gt[o:o + n]).to(DEV)),
                        proposal_boxes=Boxes(n)))
        o += n
    return out


@pytest.mark.parametrize("name", ["a", "b", "agn"])
def test_mask_head_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "mask_head.npz"))
    x, cls, gt, per_img = g[f"{name}_logits"], g[f"{name}_classes"], g[f"{name}_gt"], g[f"{name}_per_img"].tolist()
    xt = torch.from_numpy(x).to(DEV).requires_grad_(True)
    rec = Recorder()
    loss = mask_rcnn_loss(xt, _instances(cls, gt, per_img), storage=rec)
    assert loss.dtype == torch.float32 and loss.dim() == 0
    assert abs(loss.item() - float(g[f"{name}_loss"])) <= 1e-5 * abs(float(g[f"{name}_loss"]))
    for k in ("accuracy", "false_positive", "false_negative"):
        assert abs(rec.scalars[f"mask_rcnn/{k}"] - float(g[f"{name}_{k}"])) < 1e-12, k
    (loss * 1.75).backward()
    want = g[f"{name}_grad_x1p75"]
    got = xt.grad.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    assert np.array_equal(got == 0, want == 0)  # planes of the other classes are exact zeros
    pred = [Inst(pred_classes=torch.from_numpy(cls[o:o + n]).to(DEV)) for o, n in zip(np.cumsum([0] + per_img[:-1]), per_img)]
    mask_rcnn_inference(torch.from_numpy(x).to(DEV), pred)
    probs = torch.cat([p.pred_masks for p in pred]).cpu().numpy()
    assert probs.shape == g[f"{name}_probs"].shape
    assert np.abs(probs - g[f"{name}_probs"]).max() <= 1e-6
    assert [tuple(p.pred_masks.shape) for p in pred] == [(n, 1) + x.shape[2:] for n in per_img]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("B,C,M", [(256, 80, 28), (3, 7, 14), (70000, 2, 2), (5, 1, 9)])
def test_mask_head_vs_oracle(dtype, tol, B, C, M):
    rng = np.random.default_rng(B * 31 + C)
    x = torch.from_numpy((rng.standard_normal((B, C, M, M)) * 4).astype(np.float32)).to(dtype)
    cls = rng.integers(0, C, B).astype(np.int64)
    gt = rng.random((B, M, M)) < 0.3
    xn = x.float().numpy()  # the values the kernel sees
    xt = x.to(DEV).requires_grad_(True)
    loss, stats = mask_rcnn_loss_from_targets(xt, torch.from_numpy(cls).to(DEV), torch.from_numpy(gt).to(DEV))
    want, st = omh.mask_rcnn_loss(xn, cls, gt)
    assert abs(loss.item() - want) <= 1e-5 * abs(want)
    assert stats.tolist() == st["counts"].tolist() + [0]
    gup = torch.tensor(0.5, device=DEV)
    (loss * gup).backward()
    gw = omh.mask_rcnn_loss_grad(xn, cls, gt, 0.5)
    gg = xt.grad.float().cpu().numpy()
    assert xt.grad.dtype == dtype
    sub = {torch.float16: 2.0 ** -25, torch.bfloat16: 0.0, torch.float32: 0.0}[dtype]  # f16 gradients this small are subnormal
    assert np.abs(gg - gw).max() <= tol * np.abs(gw).max() + sub
    other = np.ones(gg.shape, bool)
    if C > 1:
        other[np.arange(B), cls] = False
    else:
        other[:] = False
    assert not gg[other].any()  # planes of the other classes are exact zeros
    pred = [Inst(pred_classes=torch.from_numpy(cls).to(DEV))]
    mask_rcnn_inference(x.to(DEV), pred)
    assert pred[0].pred_masks.dtype == dtype
    pw = omh.mask_rcnn_inference(xn, cls)
    assert np.abs(pred[0].pred_masks.float().cpu().numpy() - pw).max() <= tol


def test_mask_head_loss_is_deterministic_and_needs_no_sync():
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((256, 80, 28, 28)).astype(np.float32)).to(DEV).bfloat16()
    cls = torch.from_numpy(rng.integers(0, 80, 256)).to(DEV)
    gt = torch.from_numpy(rng.random((256, 28, 28)) < 0.5).to(DEV)
    a = [mask_rcnn_loss_from_targets(x, cls, gt)[0] for _ in range(4)]
    assert all(torch.equal(a[0], b) for b in a[1:])


def test_mask_head_edge_cases():
    x = torch.randn(4, 3, 14, 14, device=DEV, requires_grad=True)
    # no instances at all: `pred_mask_logits.sum() * 0` (mask_head.py:69-70), differentiable
    loss = mask_rcnn_loss(x, [Inst(gt_classes=torch.zeros(0, dtype=torch.int64, device=DEV))])
    assert loss.item() == 0.0
    loss.backward()
    assert torch.equal(x.grad, torch.zeros_like(x))
    # a gt class outside [0, C): IndexError like the reference's gather (reported at the one host read)
    bad = torch.tensor([0, 1, 3, 2], device=DEV)
    gt = torch.zeros(4, 14, 14, dtype=torch.bool, device=DEV)
    with pytest.raises(IndexError):
        mask_rcnn_loss(x, [Inst(gt_classes=bad, gt_masks=Targets(gt), proposal_boxes=Boxes(4))], storage=Recorder())
    # ... and without an event storage (nothing reads the counter, no host sync): the loss is poisoned, not silent
    silent = mask_rcnn_loss(x, [Inst(gt_classes=bad, gt_masks=Targets(gt), proposal_boxes=Boxes(4))])
    assert torch.isnan(silent).item()
    ok = mask_rcnn_loss(x, [Inst(gt_classes=torch.tensor([0, 1, 2, 2], device=DEV), gt_masks=Targets(gt),
                                  proposal_boxes=Boxes(4))])
    assert torch.isfinite(ok).item()
    # inference: out-of-range class -> NaN row, others untouched; empty input
    pred = [Inst(pred_classes=bad)]
    mask_rcnn_inference(x.detach(), pred)
    pm = pred[0].pred_masks
    assert torch.isnan(pm[2]).all() and not torch.isnan(pm[[0, 1, 3]]).any()
    e = [Inst(pred_classes=torch.zeros(0, dtype=torch.int64, device=DEV))]
    mask_rcnn_inference(torch.zeros(0, 3, 14, 14, device=DEV), e)
    assert tuple(e[0].pred_masks.shape) == (0, 1, 14, 14)
    # CPU tensors: no fallback
    with pytest.raises(NotImplementedError):
        mask_rcnn_inference(torch.zeros(1, 1, 7, 7), [Inst(pred_classes=torch.zeros(1, dtype=torch.int64))])
    # float targets are thresholded at 0.5 (mask_head.py:81-86)
    t = torch.rand(4, 14, 14, device=DEV)
    l1, _ = mask_rcnn_loss_from_targets(x.detach(), torch.tensor([0, 1, 2, 0], device=DEV), t)
    l2, _ = mask_rcnn_loss_from_targets(x.detach(), torch.tensor([0, 1, 2, 0], device=DEV), t > 0.5)
    assert torch.equal(l1, l2)
