"""SURVEY 8 row g1 / north_star "GeneralizedRCNN / RetinaNet models load unchanged": the reference's OWN callers of the hot
path, unmodified, running on HIP tensors on top of this package's operator surface -- detectron2/modeling/poolers.py
(ROIPooler), proposal_generator/proposal_utils.py (find_top_rpn_proposals), roi_heads/mask_head.py (mask_rcnn_loss /
mask_rcnn_inference), structures/masks.py (BitMasks.crop_and_resize), layers/mask_ops.py (paste_masks_in_image) -- with
the names they import from `detectron2.layers` bound to `detectron2_amd.layers` (oracle/ref.py: py_callers); each is
compared with the fused entry of this package that replaces it.  The reference modules are loaded from /root/reference
or, on the GPU box, from the bytecode oracle/build_ref.py compiled from those files into oracle/_ref/py/."""
import numpy as np
import pytest
import torch

import detectron2_amd.layers as d2l
from oracle import ref

import os

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.fixture(scope="module")
def R():
    """oracle/_ref/ is git-ignored and travels with the working tree: a checkout without it must not turn this file into
    a silent skip.  Missing reference modules FAIL unless D2AMD_NO_REFERENCE=1 states that the reference tree is
    legitimately absent on this machine."""
    if not ref.have_py():
        if os.environ.get("D2AMD_NO_REFERENCE") == "1":
            pytest.skip("D2AMD_NO_REFERENCE=1: reference modules not staged")
        pytest.fail("oracle/_ref/py/*.pyc missing: run `python -m oracle.build_ref` where /root/reference exists, "
                    "or set D2AMD_NO_REFERENCE=1 where it legitimately does not")
    return ref.py_callers(d2l)


def _boxes(g, n, w, h, smin=8, smax=300):
    s = torch.exp(torch.empty(n).uniform_(np.log(smin), np.log(smax), generator=g))
    ar = torch.exp(torch.empty(n).uniform_(np.log(0.5), np.log(2.0), generator=g))
    bw, bh = s * ar.sqrt(), s / ar.sqrt()
    cx, cy = torch.empty(n).uniform_(0, w, generator=g), torch.empty(n).uniform_(0, h, generator=g)
    b = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, w)
    b[:, 1::2] = b[:, 1::2].clamp(0, h)
    return b


@pytest.mark.parametrize("dtype,out", [(torch.float32, 7), (torch.bfloat16, 14)])
def test_reference_roipooler_runs_on_this_roialign_and_equals_the_fused_pooler(R, dtype, out):
    """modeling/poolers.py:114-263 unchanged: its assign_boxes_to_levels + per-level ROIAlign + index_put_ loop on top of
    detectron2_amd.layers.ROIAlign == the fused multi-level pooler (forward bit-identical: the same kernel evaluates a
    level's rows either way; backward: per-level tile gathers summed by autograd vs one fused launch)."""
    from detectron2_amd.modeling import ROIPooler
    from detectron2_amd.structures import Boxes

    g = torch.Generator().manual_seed(3)
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [torch.randn(2, 32, 640 // s, 800 // s, generator=g).to(dtype).to(DEV).contiguous(
        memory_format=torch.channels_last) for s in (4, 8, 16, 32)]
    bl = [_boxes(g, 150, 800, 640, 8, 640).to(DEV) for _ in range(2)]
    ref_pooler = R.poolers.ROIPooler(out, scales, 0, "ROIAlignV2")
    own_pooler = ROIPooler(out, scales, 0, "ROIAlignV2")
    xr = [f.clone().requires_grad_(True) for f in feats]
    xo = [f.clone().requires_grad_(True) for f in feats]
    yr = ref_pooler(xr, [R.Boxes(b) for b in bl])
    yo = own_pooler(xo, [Boxes(b) for b in bl])
    assert yr.shape == yo.shape == (300, 32, out, out) and torch.equal(yr, yo)
    lv = R.poolers.assign_boxes_to_levels([R.Boxes(b) for b in bl], 2, 5, 224, 4)
    assert len(torch.unique(lv)) == 4  # every level is exercised
    dy = torch.randn(yr.shape, generator=g).to(dtype).to(DEV)
    yr.backward(dy)
    yo.backward(dy)
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    for a, b in zip(xr, xo):
        assert (a.grad.float() - b.grad.float()).abs().max() <= tol * b.grad.float().abs().max()


def test_reference_find_top_rpn_proposals_runs_on_this_batched_nms(R):
    """proposal_generator/proposal_utils.py:22-135 unchanged (its per-level topk, clip, nonempty, batched_nms per image)
    on HIP tensors with detectron2.layers.batched_nms = this package's; the fused path (find_top_rpn_proposals_fused, fed
    with the head outputs) keeps the same proposals in the same order."""
    from detectron2_amd.modeling import find_top_rpn_proposals_fused

    g = torch.Generator().manual_seed(5)
    sizes = [6000, 1500, 400]
    anchors, logits, deltas = [], [], []
    for l, a in enumerate(sizes):
        c = torch.rand(a, 2, generator=g) * torch.tensor([400.0, 300.0])
        wh = 24.0 * 2 ** l * torch.exp(torch.rand(a, 2, generator=g) - 0.5)
        anchors.append(torch.cat([c - wh / 2, c + wh / 2], 1).to(DEV))
        logits.append((torch.randn(2, a, generator=g) + torch.arange(a) * 1e-6).to(DEV))
        deltas.append((torch.randn(2, a, 4, generator=g) * 0.2).to(DEV))
    hw = [(300, 400)] * 2
    own = find_top_rpn_proposals_fused(anchors, logits, deltas, hw, 0.7, 1000, 300, 0.0, True)
    # the reference function takes DECODED proposals per level: decode with the reference's formula (box_regression.py:
    # 88-116, weights 1) in torch on the device
    import math

    props = []
    for a, d in zip(anchors, deltas):
        w, h = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
        cx, cy = a[:, 0] + 0.5 * w, a[:, 1] + 0.5 * h
        dw, dh = d[..., 2].clamp(max=math.log(1000.0 / 16)), d[..., 3].clamp(max=math.log(1000.0 / 16))
        pcx, pcy = d[..., 0] * w + cx, d[..., 1] * h + cy
        pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
        props.append(torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], -1))
    got = R.proposal_utils.find_top_rpn_proposals(props, logits, hw, 0.7, 1000, 300, 0.0, True)
    for a, b in zip(got, own):
        assert a.proposal_boxes.tensor.is_cuda and len(a) == len(b) > 50
        assert torch.equal(a.objectness_logits, b.objectness_logits)
        assert torch.allclose(a.proposal_boxes.tensor, b.proposal_boxes.tensor, rtol=2e-6, atol=1e-4)


def test_reference_mask_head_functions_run_on_this_roialign(R):
    """roi_heads/mask_head.py:33-158 unchanged: mask_rcnn_loss crops the ground truth with the reference's
    BitMasks.crop_and_resize (structures/masks.py:193-224), which calls detectron2.layers.roi_align.ROIAlign = this
    package's; loss, logged statistics and gradient == the fused mask_rcnn_loss; mask_rcnn_inference likewise."""
    from detectron2_amd.modeling import mask_rcnn_inference, mask_rcnn_loss
    from detectron2_amd.structures import BitMasks, Boxes

    g = torch.Generator().manual_seed(9)
    H, W, M, C = 160, 200, 28, 5
    ref_inst, own_inst, n_rows = [], [], 0
    for n in (7, 0, 12):
        b = _boxes(g, n, W, H, 12, 120).to(DEV)
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        masks = torch.stack([((xx - (bb[0] + bb[2]).item() / 2) ** 2 / max((bb[2] - bb[0]).item() / 2, 1) ** 2 +
                              (yy - (bb[1] + bb[3]).item() / 2) ** 2 / max((bb[3] - bb[1]).item() / 2, 1) ** 2) <= 1
                             for bb in b.cpu()]).to(DEV) if n else torch.zeros(0, H, W, dtype=torch.bool, device=DEV)
        cls = torch.randint(0, C, (n,), generator=g).to(DEV)
        for Inst, BoxesT, Masks, bucket in ((R.Instances, R.Boxes, R.masks.BitMasks, ref_inst),
                                            (R.Instances, Boxes, BitMasks, own_inst)):
            i = Inst((H, W))
            i.proposal_boxes, i.gt_classes, i.gt_masks = BoxesT(b), cls, Masks(masks)
            bucket.append(i)
        n_rows += n
    logits = torch.randn(n_rows, C, M, M, generator=g).to(DEV)
    lr, lo = logits.clone().requires_grad_(True), logits.clone().requires_grad_(True)
    loss_ref = R.mask_head.mask_rcnn_loss(lr, ref_inst)
    ref_stats = dict(R.events.scalars)

    class Store:
        def __init__(self):
            self.scalars = {}

        def put_scalar(self, k, v, **kw):
            self.scalars[k] = float(v)

    st = Store()
    loss_own = mask_rcnn_loss(lo, own_inst, storage=st)
    assert abs(float(loss_ref) - float(loss_own)) <= 1e-5 * abs(float(loss_ref))
    for k, v in ref_stats.items():
        assert abs(st.scalars[k] - v) <= 1e-6, k
    loss_ref.backward()
    loss_own.backward()
    assert (lr.grad - lo.grad).abs().max() <= 1e-6 * lr.grad.abs().max() + 1e-9
    # the targets the reference's loss used == this package's BitMasks.crop_and_resize, bit for bit
    for a, b in zip(ref_inst, own_inst):
        if len(a):
            assert torch.equal(a.gt_masks.crop_and_resize(a.proposal_boxes.tensor, M),
                               b.gt_masks.crop_and_resize(b.proposal_boxes.tensor, M))
    # inference
    pr = [R.Instances((H, W)) for _ in range(3)]
    po = [R.Instances((H, W)) for _ in range(3)]
    for i, n in enumerate((7, 0, 12)):
        pc = torch.randint(0, C, (n,), generator=g).to(DEV)
        pr[i].pred_classes, po[i].pred_classes = pc, pc
    R.mask_head.mask_rcnn_inference(logits, pr)
    mask_rcnn_inference(logits, po)
    for a, b in zip(pr, po):
        assert a.pred_masks.shape == b.pred_masks.shape
        assert (a.pred_masks - b.pred_masks).abs().max().item() <= 1e-6 if a.pred_masks.numel() else True


def test_reference_paste_masks_on_the_device_equals_this_kernel(R):
    """layers/mask_ops.py:74-147 unchanged, run by torch on the HIP device (its GPU formulation: grid_sample over chunks)
    == this package's paste_masks_in_image, bit for bit, bool and uint8 outputs."""
    g = torch.Generator().manual_seed(11)
    n, H, W = 40, 240, 333
    masks = torch.rand(n, 28, 28, generator=g).to(DEV)
    b = _boxes(g, n, W, H, 6, 200)
    b[0] = torch.tensor([-10.0, -5.0, 50.0, 60.0])  # partly outside the image
    b = b.to(DEV)
    for thr in (0.5, -1.0):
        want = R.mask_ops.paste_masks_in_image(masks, R.Boxes(b), (H, W), thr)
        got = d2l.paste_masks_in_image(masks, b, (H, W), thr)
        assert want.dtype == got.dtype and want.shape == got.shape == (n, H, W)
        assert torch.equal(want, got), thr
