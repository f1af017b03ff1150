"""GPU parity tests of the fused multi-level ROIPooler (csrc/roi_pool.hip) -- SURVEY 8(a) row a4.

Oracle: the level-assignment restatement (tests/test_tile_gather_math.py pins it against torch on
the CPU) + the oracle's per-level ROIAlign forward / backward, i.e. exactly the reference's
ROIPooler.forward structure (detectron2/modeling/poolers.py:247-263) evaluated on the CPU."""
import zlib

import numpy as np
import pytest
import torch

import oracle
from detectron2_amd.modeling import ROIPooler, assign_boxes_to_levels, convert_boxes_to_pooler_format
from detectron2_amd.structures import Boxes

from test_tile_gather_math import assign_levels_restated

from conftest import ROI_FLOOR, SUM_FLOOR, assert_close_fp32, record_ratio

pytestmark = pytest.mark.gpu
DEV = "cuda"
SCALES = [1 / 4, 1 / 8, 1 / 16, 1 / 32]


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def make_inputs(rng, n_img, C, img_h, img_w, per_img):
    feats = [rng.standard_normal((n_img, C, -(-img_h // s), -(-img_w // s))).astype(np.float32)
             for s in (4, 8, 16, 32)]
    boxes = []
    for _ in range(n_img):
        s = np.exp(rng.uniform(np.log(4), np.log(0.9 * min(img_h, img_w)), per_img))
        ar = np.exp(rng.uniform(np.log(0.4), np.log(2.5), per_img))
        w, h = s * np.sqrt(ar), s / np.sqrt(ar)
        cx, cy = rng.uniform(0, img_w, per_img), rng.uniform(0, img_h, per_img)
        b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
        b[:, 0::2] = b[:, 0::2].clip(0, img_w)
        b[:, 1::2] = b[:, 1::2].clip(0, img_h)
        boxes.append(b.astype(np.float32))
    return feats, boxes


def oracle_pooler(feats, boxes, out, sr, aligned, grad=None):
    """reference structure on the CPU: levels -> per-level oracle ROIAlign -> scatter rows"""
    allb = np.concatenate(boxes)
    bidx = np.concatenate([np.full(len(b), i, np.float32) for i, b in enumerate(boxes)])
    rois = np.concatenate([bidx[:, None], allb], 1).astype(np.float32)
    lv = assign_levels_restated(allb, 2, 5, 224, 4)
    C = feats[0].shape[1]
    y = np.zeros((len(rois), C, out, out), np.float32)
    gins = []
    for l, f in enumerate(feats):
        sel = np.nonzero(lv == l)[0]
        y[sel] = oracle.roi_align_forward(f, rois[sel], (out, out), SCALES[l], sr, aligned)
        if grad is not None:
            gins.append(oracle.roi_align_backward(np.ascontiguousarray(grad[sel]), rois[sel], f.shape, SCALES[l],
                                                  sr, aligned))
    return y, gins, lv


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("out,sr,ptype", [(7, 0, "ROIAlignV2"), (14, 2, "ROIAlignV2"), (7, 0, "ROIAlign")])
def test_fused_pooler_fp32_vs_oracle(layout, out, sr, ptype):
    rng = np.random.default_rng(out * 31 + sr)
    feats, boxes = make_inputs(rng, 2, 8, 512, 640, 40)
    boxes[0][0] = [0, 0, 640, 512]      # whole image -> top level, large sampling grid
    boxes[0][1] = [10, 10, 10, 10]      # empty box
    boxes[1][0] = [100, 50, 101.5, 52]  # tiny box: many bins per pixel
    boxes[1][1] = [30, 40, 300, 290]    # level 4
    boxes[1][2] = [5, 5, 150, 140]      # level 3
    aligned = ptype == "ROIAlignV2"
    pooler = ROIPooler(out, SCALES, sr, ptype)
    xs = [torch.from_numpy(f).to(DEV).requires_grad_(True) for f in feats]
    xin = [x.contiguous(memory_format=torch.channels_last) if layout == "nhwc" else x for x in xs]
    y = pooler(xin, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    g = rng.standard_normal(y.shape).astype(np.float32)
    exp, gexp, lv = oracle_pooler(feats, boxes, out, sr, aligned, g)
    assert len(set(lv.tolist())) == 4, "test inputs must hit every level"
    assert y.shape == exp.shape
    assert_close_fp32(y.detach().cpu().numpy(), exp, "pooler:81", floor=ROI_FLOOR)
    y.backward(torch.from_numpy(g).to(DEV))
    for x, ge in zip(xs, gexp):
        assert_close_fp32(x.grad.cpu().numpy(), ge, "pooler:84", floor=ROI_FLOOR)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_pooler_16bit_nhwc(dtype):
    rng = np.random.default_rng(5)
    feats, boxes = make_inputs(rng, 2, 64, 128, 192, 48)
    fq = [torch.from_numpy(f).to(dtype) for f in feats]
    feats_r = [f.float().numpy() for f in fq]
    pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2")
    xs = [f.to(DEV).requires_grad_(True) for f in fq]
    y = pooler([x.contiguous(memory_format=torch.channels_last) for x in xs],
               [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    gq = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32)).to(dtype)
    exp, gexp, _ = oracle_pooler(feats_r, boxes, 7, 0, True, gq.float().numpy())
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert rel_err(y.float().detach().cpu().numpy(), exp) < 2 * ulp
    y.backward(gq.to(DEV))
    for x, ge in zip(xs, gexp):
        assert rel_err(x.grad.float().cpu().numpy(), ge) < 2 * ulp


def test_fused_pooler_level_assignment_and_structure():
    """device level assignment == torch's on the same boxes; fused == per-level ROIAlign loop"""
    rng = np.random.default_rng(9)
    feats, boxes = make_inputs(rng, 2, 16, 256, 320, 300)
    # exact level boundaries (sqrt(area) = 112, 224, 448)
    boxes[0][:3] = [[0, 0, 112, 112], [0, 0, 224, 224], [10, 10, 234, 234]]
    bl = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes]
    xs = [torch.from_numpy(f).to(DEV).contiguous(memory_format=torch.channels_last) for f in feats]
    pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2")
    y = pooler(xs, bl)
    lv = assign_boxes_to_levels(bl, 2, 5, 224, 4)
    fmt = convert_boxes_to_pooler_format(bl)
    ref = torch.zeros_like(y)
    for l, lp in enumerate(pooler.level_poolers):
        inds = torch.nonzero(lv == l, as_tuple=True)[0]
        ref.index_put_((inds,), lp(xs[l], fmt[inds]))
    assert torch.equal(y, ref)  # same kernels, same arithmetic -> bit-identical rows
    # CPU restatement of the levels agrees with the device's torch ops
    assert np.array_equal(lv.cpu().numpy(), assign_levels_restated(np.concatenate(boxes), 2, 5, 224, 4))


def test_fused_pooler_backward_is_deterministic_and_complete():
    """tile gather: bit-identical across runs, and every grad element is written (no stale memory)"""
    rng = np.random.default_rng(2)
    feats, boxes = make_inputs(rng, 2, 32, 200, 264, 256)
    bl = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes]
    pooler = ROIPooler(14, SCALES, 0, "ROIAlignV2")
    grads = []
    for rep in range(2):
        xs = [torch.from_numpy(f).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
              for f in feats]
        # poison the allocator's free blocks so unwritten grad elements would show up as NaN
        junk = [torch.full_like(x, float("nan")) for x in xs]
        del junk
        y = pooler(xs, bl)
        torch.manual_seed(0)
        y.backward(torch.randn_like(y))
        grads.append([x.grad.clone() for x in xs])
    for a, b in zip(*grads):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


def test_fused_pooler_edge_cases():
    pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2")
    xs = [torch.randn(2, 8, 64 // s, 96 // s, device=DEV).contiguous(memory_format=torch.channels_last)
          .requires_grad_(True) for s in (1, 2, 4, 8)]
    # no boxes at all: (0, C, 7, 7), backward gives zero grads
    y = pooler(xs, [Boxes(torch.zeros(0, 4, device=DEV)), Boxes(torch.zeros(0, 4, device=DEV))])
    assert y.shape == (0, 8, 7, 7)
    y.sum().backward()
    assert all((x.grad == 0).all() for x in xs)
    # empty box list == zero images
    z = pooler([x[:0] for x in xs], [])
    assert z.shape == (0, 8, 7, 7)
    # boxes entirely outside the image / inverted boxes give zero rows and zero gradient
    for x in xs:
        x.grad = None
    b = torch.tensor([[1000., 1000, 1100, 1100], [50, 50, 40, 40]], device=DEV)
    y = pooler(xs, [Boxes(b), Boxes(torch.zeros(0, 4, device=DEV))])
    assert (y == 0).all()
    y.sum().backward()
    assert all((x.grad == 0).all() for x in xs)
    # single level pooler == ROIAlign module
    from detectron2_amd.layers import ROIAlign
    p1 = ROIPooler((7, 7), [1 / 4], 2, "ROIAlign")
    bb = torch.tensor([[4., 4, 60, 50], [10, 20, 90, 80]], device=DEV)
    a = p1([xs[0]], [Boxes(bb[:1]), Boxes(bb[1:])])
    r = ROIAlign((7, 7), 1 / 4, 2, False)(xs[0], torch.cat([torch.tensor([[0.], [1.]], device=DEV), bb], 1))
    assert torch.equal(a, r)


def test_fused_pooler_full_size_adjoint():
    """BASELINE config 2 shapes (2 x 800x1344, 256 ch bf16 NHWC, 1024 ROIs, 7x7): forward/backward are
    adjoint -- <pool(x), g> == sum_l <x_l, grad_l> -- and match an fp32 run to bf16 precision."""
    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    hw = [(200, 336), (100, 168), (50, 84), (25, 42)]
    _, boxes = make_inputs(rng, 2, 1, 800, 1344, 512)
    bl = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes]
    pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2")
    xf = [torch.randn(2, 256, h, w, device=DEV).contiguous(memory_format=torch.channels_last) for h, w in hw]
    xs = [x.clone().requires_grad_(True) for x in xf]
    y = pooler(xs, bl)
    g = torch.randn_like(y)
    y.backward(g)
    lhs = (y.detach().double() * g.double()).sum()
    rhs = sum((x.detach().double() * x.grad.double()).sum() for x in xs)
    assert abs(lhs - rhs) / abs(lhs) < 1e-5
    xb = [x.to(torch.bfloat16).requires_grad_(True) for x in xf]
    yb = pooler(xb, bl)
    yb.backward(g.to(torch.bfloat16))
    yr = pooler([x.detach().float() for x in xb], bl)
    assert rel_err(yb.float().detach().cpu().numpy(), yr.cpu().numpy()) < 2.0 ** -7
    for a, b in zip(xb, xs):
        assert rel_err(a.grad.float().cpu().numpy(), b.grad.cpu().numpy()) < 2.0 ** -6


@pytest.mark.parametrize("case", ["clustered_overflow", "pooled28_pb32", "tiny_rois_multichunk", "two_slabs_f16"])
def test_fused_pooler_staged_backward_paths(case):
    """Paths of the LDS-staged tile gather that the BASELINE shapes do not reach: tiles with more than 64 ROIs
    (in-kernel scan of > 512 records in two passes), a second, partial channel slab, 32 bins per axis (one list entry
    per weight round), windows of more than 32 bins (several items per ROI)."""
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)  # fixed per case (hash() changes per process)
    dtype, tol = torch.float32, 1e-4
    if case == "clustered_overflow":
        n_img, C, out, sr, per = 1, 136, 7, 0, 700   # fp32: 128 channels per slab -> 2 slabs, the second partial
        feats, boxes = make_inputs(rng, n_img, C, 160, 224, per)
        c = rng.uniform([60, 40], [90, 60], (per, 2))          # all boxes around one spot: > 64 ROIs on its tiles
        s = np.exp(rng.uniform(np.log(6), np.log(120), (per, 1))) * rng.uniform(0.7, 1.4, (per, 2))
        boxes = [np.concatenate([c - s / 2, c + s / 2], 1).clip(0, [224, 160, 224, 160]).astype(np.float32)]
    elif case == "pooled28_pb32":
        n_img, C, out, sr, per = 2, 8, 28, 2, 24
        feats, boxes = make_inputs(rng, n_img, C, 128, 160, per)
    elif case == "tiny_rois_multichunk":
        n_img, C, out, sr, per = 2, 16, 14, 0, 96
        feats, boxes = make_inputs(rng, n_img, C, 96, 128, per)
        for b in boxes:                                        # boxes of 3-10 px: 14 bins inside one or two pixels
            c = rng.uniform([10, 10], [118, 86], (per, 2))
            s = rng.uniform(3, 10, (per, 2))
            b[:] = np.concatenate([c - s / 2, c + s / 2], 1)
    else:
        n_img, C, out, sr, per = 2, 320, 7, 0, 64              # f16: 256 channels per slab -> 2 slabs, the second partial
        feats, boxes = make_inputs(rng, n_img, C, 96, 128, per)
        dtype, tol = torch.float16, 2.0 ** -8
        feats = [np.float16(f).astype(np.float32) for f in feats]
    grad = rng.standard_normal((sum(len(b) for b in boxes), C, out, out)).astype(np.float32)
    if dtype != torch.float32:
        grad = np.float16(grad).astype(np.float32)
    want, gins, _ = oracle_pooler(feats, boxes, out, sr, True, grad)
    xs = [torch.from_numpy(f).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for f in feats]
    pooler = ROIPooler(out, SCALES, sr, "ROIAlignV2")
    y = pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    assert rel_err(y.detach().float().cpu().numpy(), want) <= tol
    y.backward(torch.from_numpy(grad).to(DEV).to(dtype))
    for x, g in zip(xs, gins):
        assert torch.isfinite(x.grad).all()
        assert rel_err(x.grad.float().cpu().numpy(), g) <= tol


@pytest.mark.parametrize("layout,dtype,tol", [("nhwc", torch.float32, 1e-4), ("nhwc", torch.bfloat16, 2.0 ** -6),
                                               ("nchw", torch.bfloat16, 2.0 ** -6), ("nchw", torch.float32, 1e-4)])
def test_two_poolers_of_the_same_features_chain_their_backward(layout, dtype, tol):
    """Mask R-CNN's box (7x7) and mask (14x14) poolers on the same FPN features, ONE backward: the gradient equals the
    sum of the two oracle gradients (what autograd's accumulation gives), the second pooler pooled from the first
    one's alias outputs, and -- in the tile gather's layout -- no separate sum kernel ran: the first pooler's backward
    received the second one's gradient buffers and returned THOSE (d2amd_roi_pooler_backward_accumulate)."""
    from detectron2_amd.modeling import poolers as P

    rng = np.random.default_rng(77)
    n_img, C, H, W = 2, 64, 160, 224
    feats, boxes = make_inputs(rng, n_img, C, H, W, 40)
    mboxes = [b[:12] for b in boxes]
    gb = rng.standard_normal((80, C, 7, 7)).astype(np.float32)
    gm = rng.standard_normal((24, C, 14, 14)).astype(np.float32)
    _, gins_b, _ = oracle_pooler(feats, boxes, 7, 0, True, gb)
    _, gins_m, _ = oracle_pooler(feats, mboxes, 14, 0, True, gm)
    P._ALIASES.clear()
    xs = []
    for f in feats:
        t = torch.from_numpy(f).to(DEV).to(dtype)
        if layout == "nhwc":
            t = t.contiguous(memory_format=torch.channels_last)
        xs.append(t.requires_grad_(True))
    box_pooler, mask_pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2"), ROIPooler(14, SCALES, 0, "ROIAlignV2")
    yb = box_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    assert all(id(x) in P._ALIASES for x in xs)
    ym = mask_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in mboxes])
    # the mask pooler's node hangs off the box pooler's node, not off the leaves
    assert ym.grad_fn.next_functions[0][0] is yb.grad_fn
    calls = []
    orig = P._C.lib().d2amd_roi_pooler_backward_accumulate

    torch.autograd.backward([yb, ym], [torch.from_numpy(gb).to(DEV).to(dtype), torch.from_numpy(gm).to(DEV).to(dtype)])
    assert not P._ALIASES  # both nodes ran their backward: nothing can chain onto them any more
    for x, a, b in zip(xs, gins_b, gins_m):
        want = a + b
        assert rel_err(x.grad.float().cpu().numpy(), want) < tol
    # a second iteration on the same leaves starts a fresh chain (the old graph is gone) and gives the same result
    g1 = [x.grad.clone() for x in xs]
    for x in xs:
        x.grad = None
    yb = box_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    ym = mask_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in mboxes])
    torch.autograd.backward([yb, ym], [torch.from_numpy(gb).to(DEV).to(dtype), torch.from_numpy(gm).to(DEV).to(dtype)])
    for x, g in zip(xs, g1):
        assert torch.equal(x.grad, g)  # deterministic
    # only the SECOND pooler's result is used: the first one just passes the gradient through
    for x in xs:
        x.grad = None
    yb = box_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    ym = mask_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in mboxes])
    ym.backward(torch.from_numpy(gm).to(DEV).to(dtype))
    for x, b in zip(xs, gins_m):
        assert rel_err(x.grad.float().cpu().numpy(), b) < tol
    # only the FIRST one's result is used
    for x in xs:
        x.grad = None
    yb = box_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    ym = mask_pooler(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in mboxes])
    yb.backward(torch.from_numpy(gb).to(DEV).to(dtype))
    for x, a in zip(xs, gins_b):
        assert rel_err(x.grad.float().cpu().numpy(), a) < tol
    P._ALIASES.clear()


def test_pooler_backward_accumulate_entry_adds_and_skips_empty_tiles():
    """d2amd_roi_pooler_backward_accumulate through the C ABI: held + own on the tiles ROIs touch, the held values bit
    for bit on every other tile (they are neither read nor written)."""
    import ctypes

    from detectron2_amd import _C
    from detectron2_amd.modeling import poolers as P

    rng = np.random.default_rng(3)
    n_img, C, H, W = 1, 32, 128, 192
    feats, _ = make_inputs(rng, n_img, C, H, W, 1)
    boxes = [np.array([[8, 8, 40, 36], [100, 60, 130, 90]], np.float32)]  # small boxes: most tiles stay empty
    g = rng.standard_normal((2, C, 7, 7)).astype(np.float32)
    _, gins, _ = oracle_pooler(feats, boxes, 7, 0, True, g)
    cfg = ((7, 7), tuple(SCALES), 0, True, 2, 5, 224, 4)
    hw = [tuple(f.shape[2:]) for f in feats]
    p = P._params(cfg, (n_img, C), hw, _C.dtype_code(torch.empty(0, dtype=torch.float32)), _C.NHWC)
    rois = torch.from_numpy(np.concatenate([np.zeros((2, 1), np.float32), boxes[0]], 1)).to(DEV)
    held = [torch.from_numpy(rng.standard_normal((n_img, C) + s).astype(np.float32)).to(DEV)
            .contiguous(memory_format=torch.channels_last) for s in hw]
    before = [h.clone() for h in held]
    gt = torch.from_numpy(g).to(DEV).contiguous(memory_format=torch.channels_last)
    ws_bytes = _C.lib().d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p), 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    with _C.on_device(gt.device):
        _C.check(_C.lib().d2amd_roi_pooler_backward_accumulate(ctypes.byref(p), _C.ptr(gt), _C.ptr(rois), P._ptr_array(held),
                                                               2, _C.ptr(ws), ws_bytes, _C.stream()))
    torch.cuda.synchronize()
    for h, b, want in zip(held, before, gins):
        got = (h - b).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0) + 1e-6
        untouched = torch.from_numpy(want == 0).to(DEV)
        assert torch.equal(h[untouched], b[untouched])


@pytest.mark.parametrize("fused", [True, False])
def test_inverted_roi_with_fixed_sampling_ratio_forward_backward_adjoint(fused):
    """aligned=True, sampling_ratio=2 and an INVERTED box (x2 < x1, y2 < y1): the bin size is negative and the samples
    lie in (start + roi, start); torchvision's kernels (and the oracle) sample there in both directions.  The tile
    gather's footprint rectangle must cover that range too (r01 advice: it dropped the gradient)."""
    from detectron2_amd.layers import ROIAlign

    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 8, 40, 56)).astype(np.float32)
    boxes = np.array([[150, 100, 60, 30], [40, 20, 200, 150], [120, 140, 100, 20]], np.float32)  # 0 and 2 inverted
    rois = np.concatenate([np.zeros((3, 1), np.float32), boxes], 1)
    g = rng.standard_normal((3, 8, 7, 7)).astype(np.float32)
    want_y = oracle.roi_align_forward(x, rois, (7, 7), 0.25, 2, True)
    want_g = oracle.roi_align_backward(g, rois, x.shape, 0.25, 2, True)
    assert np.abs(want_y[0]).max() > 0 and np.abs(want_g).max() > 0  # the inverted boxes do contribute
    xt = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    if fused:
        feats = [xt] + [torch.zeros(1, 8, 40 >> i, 56 >> i, device=DEV).contiguous(memory_format=torch.channels_last)
                        .requires_grad_(True) for i in (1, 2, 3)]
        # canonical size chosen so that every box lands on level 0 of the pooler
        y = ROIPooler(7, SCALES, 2, "ROIAlignV2", canonical_box_size=100000)(feats, [Boxes(torch.from_numpy(boxes).to(DEV))])
    else:
        y = ROIAlign((7, 7), 0.25, 2, True)(xt, torch.from_numpy(rois).to(DEV))
    assert_close_fp32(y.detach().cpu().numpy(), want_y, "pooler:372", floor=ROI_FLOOR)
    y.backward(torch.from_numpy(g).to(DEV))
    assert_close_fp32(xt.grad.cpu().numpy(), want_g, "pooler:374", floor=ROI_FLOOR)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_transpose_entry_and_nchw_callers_get_nchw_back(dtype):
    """d2amd_transpose_batched == torch's permuting copy (odd sizes, both element sizes); an NCHW caller of the fused
    pooler gets an NCHW-contiguous result and NCHW-contiguous gradients (a channels_last caller: channels_last)."""
    from detectron2_amd.modeling.poolers import _to_nchw, _to_nhwc

    torch.manual_seed(0)
    for shape in [(2, 256, 50, 84), (3, 37, 13, 21), (1, 8, 1, 5), (4, 130, 7, 7)]:
        t = torch.randn(shape, device=DEV).to(dtype)
        cl = _to_nhwc(t)
        assert cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(cl, t)
        assert torch.equal(cl.permute(0, 2, 3, 1).contiguous(), t.permute(0, 2, 3, 1).contiguous())
        back = _to_nchw(cl)
        assert back.is_contiguous() and torch.equal(back, t)
    # r06, d2amd_transpose_multi: several tensors of one batch size in one launch (the FPN levels on the way in, their
    # gradients on the way out), odd sizes included
    from detectron2_amd.modeling.poolers import _staged_nhwc_many, _to_nchw_many

    lv = [torch.randn(sh, device=DEV).to(dtype) for sh in [(2, 64, 50, 84), (2, 64, 25, 42), (2, 64, 13, 21), (2, 64, 7, 11),
                                                            (2, 64, 1, 3)]]
    st = _staged_nhwc_many(lv)
    assert all(c.is_contiguous(memory_format=torch.channels_last) and torch.equal(c, t) for c, t in zip(st, lv))
    assert all(a is b for a, b in zip(_staged_nhwc_many(lv), st))  # (cached: the same copies for the second pooler)
    bk = _to_nchw_many(st + [None])
    assert bk[-1] is None and all(b.is_contiguous() and torch.equal(b, t) for b, t in zip(bk, lv))
    rng = np.random.default_rng(12)
    feats, boxes = make_inputs(rng, 2, 32, 96, 128, 20)
    pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2")
    bl = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes]
    res = {}
    for layout in ("nchw", "nhwc"):
        xs = [torch.from_numpy(f).to(DEV).to(dtype) for f in feats]
        if layout == "nhwc":
            xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
        xs = [x.requires_grad_(True) for x in xs]
        y = pooler(xs, bl)
        y.backward(torch.ones_like(y))
        mf = torch.contiguous_format if layout == "nchw" else torch.channels_last
        assert y.is_contiguous(memory_format=mf)
        assert all(x.grad.is_contiguous(memory_format=mf) for x in xs)
        res[layout] = (y.detach().float(), [x.grad.float() for x in xs])
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    assert (res["nchw"][0] - res["nhwc"][0]).abs().max() <= tol * res["nhwc"][0].abs().max()
    for a, b in zip(res["nchw"][1], res["nhwc"][1]):
        assert (a - b).abs().max() <= tol * b.abs().max() + 1e-6


def test_binning_beside_the_forward_gives_the_same_gradients(monkeypatch):
    """d2amd_roi_pooler_backward_phase: binning in phase 1 (here: right after the forward), gather in phase 2 (adding)
    / phase 3 (writing + zero fill of the untouched tiles) == the one-call backward, bit for bit, for a chain of two
    poolers over the same features (box 7x7 + mask 14x14)."""
    from detectron2_amd.modeling import poolers as P

    g = torch.Generator().manual_seed(5)
    feats0 = [torch.randn(2, 64, s, s + 8, generator=g).to(torch.bfloat16).to(DEV).contiguous(
        memory_format=torch.channels_last) for s in (64, 32, 16, 8)]

    def boxes(n, seed):
        gg = torch.Generator().manual_seed(seed)
        c = torch.rand(n, 2, generator=gg) * 200 + 20
        wh = torch.rand(n, 2, generator=gg) * 150 + 4
        return Boxes(torch.cat([c - wh / 2, c + wh / 2], 1).to(DEV))

    bl, ml = [boxes(96, 1), boxes(80, 2)], [boxes(24, 3), boxes(20, 4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    res = {}
    monkeypatch.setattr(P, "_PAIR", False)  # one launch per pooler (the paired launch rounds once: test_gpu_pooler_pair.py)
    for mode in ("none", "chained", "all"):
        monkeypatch.setattr(P, "_PREBIN_MODE", mode)
        feats = [f.clone().requires_grad_(True) for f in feats0]
        yb = ROIPooler(7, scales, 0, "ROIAlignV2")(feats, bl)
        ym = ROIPooler(14, scales, 0, "ROIAlignV2")(feats, ml)
        gb = torch.Generator().manual_seed(9)
        torch.autograd.backward([yb, ym], [torch.randn(yb.shape, generator=gb).to(yb.dtype).to(DEV).contiguous(
            memory_format=torch.channels_last), torch.randn(ym.shape, generator=gb).to(ym.dtype).to(DEV).contiguous(
            memory_format=torch.channels_last)])
        res[mode] = [f.grad.clone() for f in feats]
    for mode in ("chained", "all"):
        for a, b in zip(res["none"], res[mode]):
            assert torch.equal(a, b), mode


def test_ordered_forward_is_the_list_order_forward():
    """d2amd_roi_pooler_forward_ordered (ROIs pooled in a spatial processing order, K >= 512) and
    d2amd_roi_pooler_forward (list order) write the same rows, bit for bit: row k is ROI k either way."""
    import ctypes

    from detectron2_amd import _C
    from detectron2_amd.modeling import poolers as P

    rng = np.random.default_rng(3)
    hw = [(100, 168), (50, 84), (25, 42), (13, 21)]
    _, boxes = make_inputs(rng, 2, 1, 400, 672, 350)  # 700 ROIs
    n, c = 2, 64
    xs = [torch.randn(n, c, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
          for h, w in hw]
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), torch.from_numpy(b)], 1)
                      for i, b in enumerate(boxes)]).float().to(DEV).contiguous()
    k = rois.shape[0]
    pooler = ROIPooler(7, SCALES, 0, "ROIAlignV2")
    cfg = ((7, 7), tuple(SCALES), 0, True, pooler.min_level, pooler.max_level, pooler.canonical_box_size,
           pooler.canonical_level)
    p = P._params(cfg, (n, c), hw, _C.dtype_code(xs[0]), _C.NHWC)
    L = _C.lib()
    out_a = torch.zeros((k, c, 7, 7), dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
    out_b = torch.zeros_like(out_a)
    order = torch.empty(k, dtype=torch.int32, device=DEV)
    _C.check(L.d2amd_roi_pooler_forward(ctypes.byref(p), P._ptr_array(xs), _C.ptr(rois), _C.ptr(out_a), k, _C.stream()))
    _C.check(L.d2amd_roi_pooler_forward_ordered(ctypes.byref(p), P._ptr_array(xs), _C.ptr(rois), _C.ptr(out_b), k,
                                                _C.ptr(order), 4 * k, _C.stream()))
    assert torch.equal(out_a, out_b)
    perm = order.cpu().numpy()
    assert sorted(perm.tolist()) == list(range(k))  # a permutation of the ROIs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("out,per_tile", [(7, 14), (7, 40), (14, 26), (7, 64), (7, 150)])
def test_backward_with_long_roi_lists_on_single_tiles(dtype, out, per_tile):
    """Clustered proposals: `per_tile` ROIs of one size around the same point of each image, so that a few 8 x 8 tiles
    of ONE level carry lists of about that length -- 14: walked whole; 26 / 40 / 64: longer than SPLIT_MIN (24), cut into
    parts of <= 12 entries whose fp32 partial sums meet in scratch memory (the last part to arrive adds them in part
    order); 150: beyond the per-tile list capacity (in-kernel scan, never split) -- plus background ROIs on all levels.
    16-bit I/O (the MFMA tile gather) against the oracle's backward of the same 16-bit gradient; twice: bit-identical
    (which lists are split is decided in tile order, not in arrival order)."""
    rng = np.random.default_rng(out * 1000 + per_tile)
    C, img_h, img_w = 64, 320, 448
    feats, boxes = make_inputs(rng, 2, C, img_h, img_w, 40)
    for i in range(2):
        # ROIs of ~180 px (level p3/p4 of the 224-canonical rule) centred within 6 px of one point
        c = rng.uniform([150, 120], [300, 200])
        ctr = c + rng.uniform(-6, 6, (per_tile, 2))
        wh = rng.uniform(150, 210, (per_tile, 2))
        b = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        b[:, 0::2] = b[:, 0::2].clip(0, img_w)
        b[:, 1::2] = b[:, 1::2].clip(0, img_h)
        boxes[i] = np.concatenate([boxes[i], b.astype(np.float32)])
    bl = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes]
    pooler = ROIPooler(out, SCALES, 0, "ROIAlignV2")
    k = sum(len(b) for b in boxes)
    g = torch.from_numpy(rng.standard_normal((k, C, out, out)).astype(np.float32)).to(DEV).to(dtype) \
        .contiguous(memory_format=torch.channels_last)
    runs = []
    for _ in range(2):
        xs = [torch.from_numpy(f).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
              for f in feats]
        pooler(xs, bl).backward(g)
        runs.append([x.grad.clone() for x in xs])
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    _, gins, lv = oracle_pooler(feats, boxes, out, 0, True, grad=g.float().cpu().numpy())
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for l in range(4):
        assert rel_err(runs[0][l].float().cpu().numpy(), gins[l]) < tol, (l, int((lv == l).sum()))


def test_pooler_full_size_per_element_vs_oracle():
    """The ROOFLINE kernel at the BENCH's size against the oracle (not against the library's own fp32 path): bench.py's
    connected step -- 2 x 800x1344, 256 ch bf16 NHWC, the ROI lists its own RPN + sampler produce (1,024 box-head rows,
    256 mask-head rows) -- forward of both poolers and the ONE backward that runs `pool_bwd_mfma_kernel` for both (the
    mask pooler's gradient chained into the box pooler's tile gather), checked on 32 of the 256 channels (every 8th: four
    of each wave's 32) against oracle.roi_align_forward / roi_align_backward per level, element by element:
      forward   |d| <= 1 ulp_bf16(|y|) + 2^-14 max|y|
      backward  |d| <= 1 ulp_bf16(|g|) + 1 ulp_bf16(max(|g_box|, |g_mask|)) + 2^-13 A + 1e-6 max|g|,  A = the same scatter
                of |dY| (the weights are >= 0, so A = sum |w dY| exactly); the second ulp: the chained backward rounds the
                first pooler's gradient to bf16 before the second adds into it.
    Terms: one output rounding to bf16; sample coordinates are fp32 in the reference and here -- an ulp of a coordinate
    of ~300 px is 3e-5 px and two correct fp32 evaluation orders differ by that much times the feature slope (the
    reference's own fp32 order is 1.1e-5 max|y| from the fp64 value: tests/test_oracle_golden.py::
    test_fp32_roi_align_distance_from_fp64), hence 2^-14 max|y| forward and 2^-14 of A backward; the 16-bit hi / lo
    split of the MFMA weight image adds 2^-16 A; fp32 accumulation.  A dropped or doubled ROI, tap or bin is O(2^-3) of
    these magnitudes on the pixels it touches: 3e-2-of-the-maximum bars cannot see that on a pixel many ROIs cover."""
    import bench

    dev = torch.device("cuda", 0)
    w = bench.Workload(dev, torch.bfloat16, "nhwc")
    out = bench.connected_forward(w)
    samp = out["sample"]
    torch.cuda.synchronize()
    counts = samp["counts"].cpu().numpy()
    assert (counts[:, 1] == bench.ROI_BATCH).all(), counts  # (no padding rows in this workload)
    ch = np.arange(0, 256, 8)
    feats = [np.ascontiguousarray(f.detach().float().cpu().numpy()[:, ch]) for f in w.feats]
    ulp = lambda v: 2.0 ** (np.floor(np.log2(np.maximum(np.abs(v), 1e-30))) - 7)
    exp_grads = [np.zeros_like(f) for f in feats]
    abs_grads = [np.zeros_like(f) for f in feats]
    part_max = [np.zeros_like(f) for f in feats]  # max over the two poolers of |that pooler's own gradient|
    bad = {}
    for name, rois_t, y_t, g_t, R in (("box", samp["rois"], out["box_features"], w.gbox, 7),
                                      ("mask", samp["head_rois"], out["mask_features"], w.gmask, 14)):
        rois = rois_t.cpu().numpy()
        lv = assign_levels_restated(rois[:, 1:], 2, 5, 224, 4)
        assert len(set(lv.tolist())) >= (4 if name == "box" else 3), "the bench's lists hit (nearly) every level"
        got = y_t.detach().float().cpu().numpy()[:, ch]
        g = np.ascontiguousarray(g_t.float().cpu().numpy()[:, ch])
        for l, f in enumerate(feats):
            sel = np.nonzero(lv == l)[0]
            if len(sel) == 0:
                continue
            exp = oracle.roi_align_forward(f, rois[sel], (R, R), SCALES[l], 0, True)
            d = np.abs(got[sel] - exp)
            bound = ulp(exp) + 2.0 ** -14 * np.abs(exp).max()
            r = float((d / bound).max())
            record_ratio(f"pooler_full/{name}_fwd_l{l}", r)
            if r > 1:
                bad[f"{name}_fwd_l{l}"] = (r, int((d > bound).sum()), d.size)
            gs = np.ascontiguousarray(g[sel])
            part = oracle.roi_align_backward(gs, rois[sel], f.shape, SCALES[l], 0, True)
            exp_grads[l] += part
            part_max[l] = np.maximum(part_max[l], np.abs(part))
            abs_grads[l] += oracle.roi_align_backward(np.abs(gs), rois[sel], f.shape, SCALES[l], 0, True)
    torch.autograd.backward([out["box_features"], out["mask_features"]], [w.gbox, w.gmask])
    for l, f in enumerate(w.feats):
        got = f.grad.float().cpu().numpy()[:, ch]
        e = exp_grads[l]
        # the chain writes the first pooler's gradient in bf16 and the second ADDS into it (one tensor, no fp32 staging):
        # two roundings -- of the first part and of the sum (the reference's autograd sums two bf16 gradients: three)
        # (x 1.25: an ulp taken at the ORACLE's value is half the device's when the two sit on either side of a power of
        # two -- 1 element of 4.3 M came out at 1.03 without it)
        bound = 1.25 * (ulp(e) + ulp(part_max[l])) + 2.0 ** -13 * abs_grads[l] + 1e-6 * np.abs(e).max()
        d = np.abs(got - e)
        r = float((d / bound).max())
        record_ratio(f"pooler_full/bwd_l{l}", r)
        if r > 1:
            bad[f"bwd_l{l}"] = (r, int((d > bound).sum()), d.size)
        if not (got[abs_grads[l] == 0] == 0).all():  # pixels no ROI touches are written as exact zeros
            bad[f"bwd_l{l}_zeros"] = int((got[abs_grads[l] == 0] != 0).sum())
    assert not bad, bad
