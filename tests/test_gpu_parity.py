"""GPU parity tests: the HIP path (through the C ABI, via the detectron2.layers-style surface)
against the CPU oracle on the same seeded inputs, against the committed golden fixtures, and
through size-independent properties at BASELINE.json's full sizes.

Bars (BASELINE.json north_star): bit-exact for NMS / IoU / indexing / paste; <= 1e-4 relative
for ROIAlign / DCN floating point (fp32 I/O); 16-bit I/O is checked against the oracle evaluated
on the same rounded inputs with a tolerance of a few 16-bit ulps."""
import os

import numpy as np
import pytest
import torch

import oracle
from detectron2_amd import layers, structures
from detectron2_amd.layers import (DeformConv, ModulatedDeformConv, ROIAlign, ROIAlignRotated, batched_nms,
                                   batched_nms_rotated, nms, nms_rotated, paste_masks_in_image,
                                   pairwise_iou_rotated)
from detectron2_amd.structures import Boxes, pairwise_intersection, pairwise_ioa, pairwise_iou

from _torch_ref import dcn_torch
from conftest import ROI_FLOOR, SUM_FLOOR, assert_close_fp32

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(DEV).to(dtype)


def random_rois(rng, k, n_img, W, H, scale, min_size=1.0, max_size=None):
    """xyxy boxes in image coordinates (feature size / scale)."""
    iw, ih = W / scale, H / scale
    max_size = max_size or 0.6 * min(iw, ih)
    s = np.exp(rng.uniform(np.log(min_size), np.log(max_size), k))
    ar = np.exp(rng.uniform(np.log(0.5), np.log(2.0), k))
    w, h = s * np.sqrt(ar), s / np.sqrt(ar)
    cx, cy = rng.uniform(0, iw, k), rng.uniform(0, ih, k)
    b = np.stack([rng.integers(0, n_img, k), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    return b.astype(np.float32)


# ======================================================================== ROIAlign
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("out,sr,aligned", [((7, 7), 0, True), ((14, 14), 2, True), ((7, 7), 0, False),
                                            ((5, 3), 3, True)])
def test_roi_align_forward_backward_fp32(layout, out, sr, aligned):
    rng = np.random.default_rng(10)
    N, C, H, W, K = 2, 8, 25, 31, 60
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = random_rois(rng, K, N, W, H, 0.25, 2.0)
    rois[0, 1:] = [-20, -10, 30, 40]      # partly outside
    rois[1, 1:] = [50, 50, 50, 50]        # empty
    rois[2, 1:] = [0, 0, 4 * W, 4 * H]    # whole map (large sampling grid when sr = 0)
    rois[3, 1:] = [500, 500, 600, 600]    # fully outside
    op = ROIAlign(out, 0.25, sr, aligned)
    xt = cu(x).requires_grad_(True)
    xin = xt.contiguous(memory_format=torch.channels_last) if layout == "nhwc" else xt
    y = op(xin, cu(rois))
    exp = oracle.roi_align_forward(x, rois, out, 0.25, sr, aligned)
    assert y.shape == exp.shape
    assert_close_fp32(y.detach().cpu().numpy(), exp, "parity:67", floor=ROI_FLOOR)
    g = rng.standard_normal(exp.shape).astype(np.float32)
    y.backward(cu(g))
    gexp = oracle.roi_align_backward(g, rois, x.shape, 0.25, sr, aligned)
    assert_close_fp32(xt.grad.cpu().numpy(), gexp, "parity:71", floor=ROI_FLOOR)


def test_roi_align_known_answers_gpu():
    # /root/reference/tests/layers/test_roi_align.py:14-47
    x = torch.arange(25, dtype=torch.float32, device=DEV).reshape(1, 1, 5, 5)
    rois = torch.tensor([[0, 1, 1, 3, 3.0]], device=DEV)
    old = ROIAlign((4, 4), 1.0, 0, aligned=False)(x, rois)[0, 0].cpu().numpy()
    new = ROIAlign((4, 4), 1.0, 0, aligned=True)(x, rois)[0, 0].cpu().numpy()
    assert np.allclose(old, [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]])
    assert np.allclose(new, [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0],
                             [12.0, 12.5, 13.0, 13.5]])


def test_roi_align_empty_box_and_batch_gpu():
    # test_roi_align.py:111-128
    x = torch.rand(1, 1, 5, 5, device=DEV, requires_grad=True)
    o = ROIAlign((7, 7), 1.0, 0)(x, torch.tensor([[0, 3, 4, 5, 4.0]], device=DEV))
    assert o.shape == (1, 1, 7, 7) and (o == 0).all()
    o.sum().backward()
    assert torch.allclose(x.grad, torch.zeros_like(x))
    e = ROIAlign((7, 7), 1.0, 0)(torch.zeros(0, 3, 10, 10, device=DEV), torch.zeros(0, 5, device=DEV))
    assert e.shape == (0, 3, 7, 7)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_roi_align_16bit(dtype, layout):
    rng = np.random.default_rng(11)
    N, C, H, W, K = 2, 16, 20, 28, 40
    xq = torch.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32)).to(dtype)
    x = xq.float().numpy()
    rois = random_rois(rng, K, N, W, H, 0.25, 2.0)
    xt = xq.to(DEV).requires_grad_(True)
    xin = xt.contiguous(memory_format=torch.channels_last) if layout == "nhwc" else xt
    y = ROIAlign((7, 7), 0.25, 0, True)(xin, cu(rois))
    assert y.dtype == dtype
    exp = oracle.roi_align_forward(x, rois, (7, 7), 0.25, 0, True)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert rel_err(y.float().detach().cpu().numpy(), exp) < 2 * ulp
    gq = torch.from_numpy(rng.standard_normal(exp.shape).astype(np.float32)).to(dtype)
    y.backward(gq.to(DEV))
    gexp = oracle.roi_align_backward(gq.float().numpy(), rois, x.shape, 0.25, 0, True)
    assert rel_err(xt.grad.float().cpu().numpy(), gexp) < 2 * ulp


def test_roi_align_rotated_golden_and_oracle(golden_dir):
    d = np.load(os.path.join(golden_dir, "roi_align_rotated.npz"))
    N, C, H, W = d["x"].shape
    for sr in (0, 2):
        xt = cu(d["x"]).requires_grad_(True)
        y = ROIAlignRotated((7, 7), 0.5, sr)(xt, cu(d["rois"]))
        assert_close_fp32(y.detach().cpu().numpy(), d[f"out_sr{sr}"], "parity:123", floor=ROI_FLOOR)
        y.backward(cu(d["grad"]))
        assert_close_fp32(xt.grad.cpu().numpy(), d[f"gin_sr{sr}"], "parity:125", floor=ROI_FLOOR)
    # channels_last + known answers test_roi_align_rotated.py:30-71
    x = torch.arange(25, dtype=torch.float32, device=DEV).reshape(1, 1, 5, 5)
    exp = np.array([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]])
    for k, ang in enumerate((0, 90, 180, 270)):
        o = ROIAlignRotated((4, 4), 1.0, 0)(x, torch.tensor([[0, 2, 2, 2, 2, float(ang)]], device=DEV))
        assert np.allclose(o[0, 0].cpu().numpy(), np.rot90(exp, -k), atol=1e-4)
    xr = np.random.default_rng(3).standard_normal((2, 8, 12, 14)).astype(np.float32)
    rr = d["rois"][:20].copy()
    a = ROIAlignRotated((7, 7), 0.5, 2)(cu(xr), cu(rr))
    b = ROIAlignRotated((7, 7), 0.5, 2)(cu(xr).contiguous(memory_format=torch.channels_last), cu(rr))
    assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
    assert_close_fp32(a.cpu().numpy(), oracle.roi_align_rotated_forward(xr, rr, (7, 7), 0.5, 2), "parity:137", floor=ROI_FLOOR)


def test_roi_align_equals_rotated_zero_angle_gpu():
    # /root/reference/tests/modeling/test_roi_pooler.py:14-59
    rng = np.random.default_rng(1)
    x = rng.random((2, 4, 10, 8), dtype=np.float32)
    b = rng.random((10, 4), dtype=np.float32) * 64
    b[:, 2:] = b[:, :2] + np.maximum(b[:, 2:], 1.0)
    bi = rng.integers(0, 2, 10).astype(np.float32)
    rois = np.concatenate([bi[:, None], b], 1)
    rrois = np.stack([bi, (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1],
                      np.zeros(10, np.float32)], 1)
    a = ROIAlign((14, 14), 1 / 16, 0, True)(cu(x), cu(rois))
    r = ROIAlignRotated((14, 14), 1 / 16, 0)(cu(x), cu(rrois))
    assert torch.allclose(a, r, atol=1e-4)


def test_roi_align_full_size_properties():
    """BASELINE config 2 shapes: p2 level, 1024 ROIs, C=256 bf16 NHWC + fp32 NCHW.  Properties:
    (1) linearity in the input, (2) constant input -> constant output inside the map,
    (3) <forward(x), g> == <x, backward(g)> (adjointness), sampled check vs oracle."""
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    N, C, H, W, K = 2, 256, 200, 336, 1024
    rois_np = random_rois(rng, K, N, W, H, 0.25, 16, 224)
    rois = cu(rois_np)
    op = ROIAlign((7, 7), 0.25, 0, True)
    x = torch.randn(N, C, H, W, device=DEV)
    y1 = op(x, rois)
    y2 = op(2.5 * x, rois)
    assert torch.allclose(y2, 2.5 * y1, rtol=1e-5, atol=1e-5)
    ones = op(torch.ones(N, C, H, W, device=DEV), rois)
    inside = (rois[:, 1] > 4) & (rois[:, 2] > 4) & (rois[:, 3] < 4 * W - 8) & (rois[:, 4] < 4 * H - 8)
    assert torch.allclose(ones[inside], torch.ones_like(ones[inside]), atol=1e-5)
    xg = x.clone().requires_grad_(True)
    y = op(xg, rois)
    g = torch.randn_like(y)
    y.backward(g)
    lhs, rhs = (y.detach().double() * g.double()).sum(), (x.double() * xg.grad.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-4
    # sampled oracle check (32 ROIs, 8 channels)
    sel = rng.choice(K, 32, replace=False)
    exp = oracle.roi_align_forward(x[:, :8].cpu().numpy(), rois_np[sel], (7, 7), 0.25, 0, True)
    assert_close_fp32(y1[sel][:, :8].cpu().numpy(), exp, "parity:181", floor=ROI_FLOOR)
    # NHWC bf16 == NCHW fp32 on the same bf16-rounded input, to bf16 precision
    xb = x.to(torch.bfloat16)
    yb = op(xb.contiguous(memory_format=torch.channels_last), rois)
    yf = op(xb.float(), rois)
    assert rel_err(yb.float().cpu().numpy(), yf.cpu().numpy()) < 2.0 ** -7


# ======================================================================== IoU
def test_pairwise_iou_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "pairwise_iou.npz"))
    B1, B2 = Boxes(cu(d["b1"])), Boxes(cu(d["b2"]))
    assert np.array_equal(pairwise_iou(B1, B2).cpu().numpy(), d["iou"])
    assert np.array_equal(pairwise_ioa(B1, B2).cpu().numpy(), d["ioa"])
    assert np.array_equal(pairwise_intersection(B1, B2).cpu().numpy(), d["intersection"])
    # known answers /root/reference/tests/structures/test_boxes.py:152-186
    b1 = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]], device=DEV)
    b2 = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.5, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 0.5, 0.5],
                       [0.5, 0.5, 1.0, 1.0], [0.5, 0.5, 1.5, 1.5]], device=DEV)
    exp = torch.tensor([[1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]] * 2)
    assert torch.allclose(pairwise_iou(Boxes(b1), Boxes(b2)).cpu(), exp)
    # odd sizes (scalar-store path), NaN rows, empty
    rng = np.random.default_rng(5)
    for n, m in [(1, 1), (3, 1001), (130, 257), (0, 5), (5, 0)]:
        a = rng.uniform(0, 50, (n, 4)).astype(np.float32); a[:, 2:] += a[:, :2]
        b = rng.uniform(0, 50, (m, 4)).astype(np.float32); b[:, 2:] += b[:, :2]
        if n > 2 and m > 2:
            a[1, 0] = np.nan; b[2, 3] = np.nan
        for mode, fn in (("iou", pairwise_iou), ("ioa", pairwise_ioa), ("intersection", pairwise_intersection)):
            got = fn(Boxes(cu(a)), Boxes(cu(b))).cpu().numpy()
            assert np.array_equal(got, oracle.pairwise_iou(a, b, mode), equal_nan=True), (n, m, mode)


def test_pairwise_iou_full_size():
    """RPN matching shape: 16 GT x 268,569 anchors, bit-exact vs oracle."""
    rng = np.random.default_rng(6)
    gt = rng.uniform(0, 800, (16, 4)).astype(np.float32); gt[:, 2:] = gt[:, :2] + rng.uniform(16, 512, (16, 2)).astype(np.float32)
    an = rng.uniform(-50, 1300, (268569, 4)).astype(np.float32); an[:, 2:] = an[:, :2] + rng.uniform(8, 700, (268569, 2)).astype(np.float32)
    got = pairwise_iou(Boxes(cu(gt)), Boxes(cu(an))).cpu().numpy()
    assert np.array_equal(got, oracle.pairwise_iou(gt, an))


def test_box_iou_rotated_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "rotated_iou_nms.npz"))
    got = pairwise_iou_rotated(cu(d["b1"]), cu(d["b2"])).cpu().numpy()
    assert got.dtype == np.float32
    nd = int((got != d["iou"]).sum())
    assert nd == 0, f"{nd} of {got.size} rotated IoUs differ, max abs {np.abs(got - d['iou']).max()}"
    # known answers /root/reference/tests/structures/test_rotated_boxes.py
    f = lambda a, b: pairwise_iou_rotated(cu(np.array(a, np.float32)), cu(np.array(b, np.float32))).cpu().numpy()
    assert np.allclose(f([[0.5, 0.5, 1, 1, 0]], [[0.25, 0.5, 0.5, 1, 0]]), 0.5)
    # 45 degrees: /root/reference/tests/structures/test_rotated_boxes.py:277-290 (both 0.5), and a square against its own
    # 45-degree turn: a regular octagon of area 8 (sqrt 2 - 1) inside two squares of area 4 -> 1 / sqrt 2
    r2 = float(np.sqrt(2))
    assert np.allclose(f([[1, 1, r2, r2, 45], [1, 1, 2 * r2, 2 * r2, -45]], [[1, 1, 2, 2, 0]]), [[0.5], [0.5]], atol=1e-6)
    assert np.allclose(f([[1, 1, 2, 2, 0]], [[1, 1, 2, 2, 45]]), 8 * (r2 - 1) / (8 - 8 * (r2 - 1)), atol=1e-5)
    assert np.allclose(f([[5, 5, 10, 6, 55]], [[5, 5, 10, 6, -35]]), 36 / 84, atol=1e-5)
    assert f([[160.0, 153.0, 230.0, 23.0, -37.0]], [[-0.122, 197.5, 0.122, 155.5, 90.0]])[0, 0] < 1e-4
    # shape with a huge M (test_rotated_boxes.py:71-76 uses 5 x 1,289,035)
    big = pairwise_iou_rotated(torch.rand(5, 5, device=DEV) * 50 + 1, torch.rand(200003, 5, device=DEV) * 50 + 1)
    assert big.shape == (5, 200003) and bool((big >= 0).all())


# ======================================================================== NMS
def _boxes(rng, n, size=100.0):
    b = rng.random((n, 4), dtype=np.float32) * np.float32(size * 0.5)
    b = np.maximum(b, 1.0)
    b[:, 2:] += b[:, :2]
    return b.astype(np.float32)


def _distinct_scores(rng, n):
    return (rng.permutation(n).astype(np.float32) + 1) / np.float32(n + 1)


# (2176 .. 2305: 34, 35, 36, 37 blocks of 64 -- the reduction's pusher waves hold a whole block of rows up to 35 blocks
# of a segment (r06) and half blocks beyond; 4097 / 4400: rows with more than 64 later words)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 1000, 2176, 2239, 2240, 2241, 2305, 4097, 4400])
def test_nms_bit_exact(n):
    rng = np.random.default_rng(100 + n)
    b, s = _boxes(rng, n), _distinct_scores(rng, n)
    for thr in (0.2, 0.5, 0.7, 0.8):
        got = nms(cu(b), cu(s), thr)
        assert got.dtype == torch.int64
        assert np.array_equal(got.cpu().numpy(), oracle.nms(b, s, thr)), (n, thr)


@pytest.mark.parametrize("n,ncls", [(1, 1), (65, 3), (1000, 5), (2000, 50), (8819, 5), (20000, 80)])
def test_batched_nms_bit_exact(n, ncls):
    rng = np.random.default_rng(200 + n)
    b, s = _boxes(rng, n, 200.0), _distinct_scores(rng, n)
    idx = rng.integers(0, ncls, n)
    bt, st, it = cu(b), cu(s), torch.from_numpy(idx).to(DEV)
    b0 = bt.clone()
    for thr in (0.5, 0.7):
        got = batched_nms(bt, st, it, thr).cpu().numpy()
        assert np.array_equal(got, oracle.batched_nms(b, s, idx, thr)), (n, ncls, thr)
    assert torch.equal(bt, b0)  # inputs not modified (test_nms.py:26-28)


def test_batched_nms_large_ties_and_wide_class_ids():
    """Beyond the brute-force ranking (n > 12,288) the order comes from the library's own stable counting sort
    (csrc/nms.hip: cs_sort): four 8-bit passes over the score bits, two over 16-bit class ids.  Tied scores must keep
    index order through every pass, and class ids above 255 exercise the second class digit."""
    rng = np.random.default_rng(1234)
    n = 20000
    b = _boxes(rng, n, 300.0)
    s = (rng.integers(0, 997, n).astype(np.float32) - 300.0) / 256.0  # ~20 ties per value, both signs, exact in fp32
    idx = rng.integers(0, 40000, n)
    idx[: n // 2] = rng.integers(0, 300, n // 2)  # populated classes on both sides of the 8-bit digit boundary
    for thr in (0.5,):
        got = batched_nms(cu(b), cu(s), torch.from_numpy(idx).to(DEV), thr).cpu().numpy()
        assert np.array_equal(got, oracle.batched_nms(b, s, idx, thr))
    got = nms(cu(b), cu(s), 0.6).cpu().numpy()
    assert np.array_equal(got, oracle.nms(b, s, 0.6))


def test_batched_nms_config4_100k():
    """BASELINE config 4: 100k candidates, 80 classes.  Bit-exact vs oracle (per-class greedy)."""
    rng = np.random.default_rng(4)
    n = 100000
    wh = np.exp(rng.uniform(np.log(8), np.log(400), (n, 2))).astype(np.float32)
    xy = (rng.random((n, 2), dtype=np.float32) * np.array([1344, 800], np.float32))
    b = np.concatenate([xy - wh / 2, xy + wh / 2], 1).astype(np.float32)
    s = _distinct_scores(rng, n)
    idx = rng.integers(0, 80, n)
    got = batched_nms(cu(b), cu(s), torch.from_numpy(idx).to(DEV), 0.5).cpu().numpy()
    exp = oracle.batched_nms(b, s, idx, 0.5)
    assert np.array_equal(got, exp)
    # properties: sorted by decreasing score, idempotent
    assert np.all(np.diff(s[got]) < 0)
    again = batched_nms(cu(b[got]), cu(s[got]), torch.from_numpy(idx[got]).to(DEV), 0.5).cpu().numpy()
    assert np.array_equal(again, np.arange(len(got)))


def test_batched_nms_images_equals_per_image_loop():
    """nms_images (one call for the batch, one host sync) == the per-image loop, bit for bit, including an empty
    image and images of different sizes; with an image beyond the batched pipeline's limit the batch takes the
    side-stream route."""
    from detectron2_amd.layers import batched_nms_images
    rng = np.random.default_rng(77)
    inputs = []
    for n in (3000, 0, 8819, 1):
        b = rng.uniform(0, 800, (n, 4)).astype(np.float32)
        b[:, 2:] = b[:, :2] + rng.uniform(4, 200, (n, 2)).astype(np.float32)
        s = ((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)
        idx = rng.integers(0, 5, n)
        inputs.append((cu(b), cu(s), torch.as_tensor(idx).to(DEV)))
    big = rng.uniform(0, 800, (13000, 4)).astype(np.float32)
    big[:, 2:] = big[:, :2] + rng.uniform(4, 200, (13000, 2)).astype(np.float32)
    big_in = (cu(big), cu(((rng.permutation(13000) + 1) / 13001).astype(np.float32)),
              torch.as_tensor(rng.integers(0, 5, 13000)).to(DEV))
    for rep in range(3):  # repeated calls reuse the side streams
        for batch in (inputs, inputs + [big_in]):
            got = batched_nms_images(batch, 0.7)
            for (b, s, i), g in zip(batch, got):
                exp = batched_nms(b, s, i, 0.7)
                assert g.dtype == torch.int64 and torch.equal(g, exp)
    b, s, i = inputs[2]
    assert np.array_equal(got[2].cpu().numpy(),
                          oracle.batched_nms(b.cpu().numpy(), s.cpu().numpy(), i.cpu().numpy(), 0.7))


def test_nms_batched_pipeline_mixed_batch():
    """d2amd_nms_batched: more images than one launch group (8), sizes from 1 to 12288 boxes, an image without
    categories, ties in the scores, degenerate (zero-area / non-finite) boxes -- each image equals the single-image
    call and the oracle."""
    from detectron2_amd.layers.ops import nms_images
    rng = np.random.default_rng(78)
    inputs = []
    for k, n in enumerate((64, 65, 1, 2000, 0, 12288, 513, 129, 8819, 700, 4097)):
        b = rng.uniform(0, 600, (n, 4)).astype(np.float32)
        b[:, 2:] = b[:, :2] + rng.uniform(0, 150, (n, 2)).astype(np.float32)
        s = rng.integers(0, 50, n).astype(np.float32) / 50  # many exact ties
        if n > 100:
            b[5] = b[6]                      # duplicates
            b[7, 2:] = b[7, :2]              # zero area
            b[9, 0] = np.inf                 # non-finite coordinate: literal formula for its tiles
            b[11, 1] = np.nan
        idx = None if k == 2 or k == 7 else torch.as_tensor(rng.integers(0, 3 + k, n)).to(DEV)
        inputs.append((cu(b), cu(s), idx))
    got = nms_images(inputs, 0.5)
    for (b, s, i), g in zip(inputs, got):
        exp = nms(b, s, 0.5) if i is None else batched_nms(b, s, i, 0.5)
        assert g.dtype == torch.int64 and torch.equal(g, exp)
    for k in (3, 6, 7, 9):
        b, s, i = inputs[k]
        ref = (oracle.nms(b.cpu().numpy(), s.cpu().numpy(), 0.5) if i is None else
               oracle.batched_nms(b.cpu().numpy(), s.cpu().numpy(), i.cpu().numpy(), 0.5))
        assert np.array_equal(got[k].cpu().numpy(), ref), k


def test_nms_threshold_boundaries_division_free():
    """The mask kernel replaces ovr > thr by an exact double-precision product test: probe thresholds that are
    floats, midpoints between floats and values just around them, on boxes built to have IoU exactly at simple
    fractions, against the oracle's literal formula; and the literal-formula switch gives the same answers."""
    rng = np.random.default_rng(79)
    n = 1500
    # integer-coordinate boxes on a small lattice: many pairs share an IoU that is an exact small fraction
    x1 = rng.integers(0, 24, n); y1 = rng.integers(0, 24, n)
    b = np.stack([x1, y1, x1 + rng.integers(1, 9, n), y1 + rng.integers(1, 9, n)], 1).astype(np.float32)
    s = _distinct_scores(rng, n)
    f = np.float32(1 / 3)
    g = np.nextafter(f, np.float32(1))
    thrs = [0.5, 0.25, float(f), float(g), (float(f) + float(g)) / 2, float(np.nextafter(f, np.float32(0))),
            1 / 3, 0.2, 1.0 / 7, 0.6000000238418579, 0.6, 2.0 / 3, 0.75, 1.0, 1e-3]
    for thr in thrs:
        got = nms(cu(b), cu(s), thr).cpu().numpy()
        assert np.array_equal(got, oracle.nms(b, s, thr)), thr


def test_nms_rotated_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "rotated_iou_nms.npz"))
    for thr in (0.2, 0.5, 0.7):
        got = nms_rotated(cu(d["nms_boxes"]), cu(d["nms_scores"]), thr).cpu().numpy()
        assert np.array_equal(got, d[f"keep_{int(thr * 10)}"]), thr
    rng = np.random.default_rng(8)
    n = 700
    b = np.stack([rng.uniform(0, 200, n), rng.uniform(0, 200, n), rng.uniform(2, 60, n), rng.uniform(2, 60, n),
                  rng.uniform(-180, 180, n)], 1).astype(np.float32)
    s = _distinct_scores(rng, n)
    idx = rng.integers(0, 4, n)
    got = batched_nms_rotated(cu(b), cu(s), torch.from_numpy(idx).to(DEV), 0.4).cpu().numpy()
    assert np.array_equal(got, oracle.batched_nms(b, s, idx, 0.4, rotated=True))
    # 0-degree rotated NMS vs horizontal NMS: test_nms_rotated.py:100-112 allows edit distance <= 1
    hb = _boxes(rng, 300)
    rb = np.stack([(hb[:, 0] + hb[:, 2]) / 2, (hb[:, 1] + hb[:, 3]) / 2, hb[:, 2] - hb[:, 0], hb[:, 3] - hb[:, 1],
                   np.zeros(300, np.float32)], 1)
    hs = _distinct_scores(rng, 300)
    k1, k2 = nms(cu(hb), cu(hs), 0.5).cpu().numpy(), nms_rotated(cu(rb), cu(hs), 0.5).cpu().numpy()
    assert len(set(k1) ^ set(k2)) <= 2


class _RotatedNmsModule(torch.nn.Module):
    """The module of the reference's scriptability test (tests/layers/test_nms_rotated.py:153-168)."""

    def forward(self, boxes, scores, threshold: float):
        return nms_rotated(boxes, scores, threshold)


def test_nms_rotated_module_is_scriptable_with_identical_results():
    """test_nms_rotated.py:161-168 (`torch.jit.script(module)` must succeed on the device), plus: scripted == eager
    == oracle."""
    m = _RotatedNmsModule().to(DEV)
    scripted = torch.jit.script(m)
    rng = np.random.default_rng(18)
    n = 300
    b = np.stack([rng.uniform(0, 120, n), rng.uniform(0, 120, n), rng.uniform(2, 50, n), rng.uniform(2, 50, n),
                  rng.uniform(-180, 180, n)], 1).astype(np.float32)
    s = _distinct_scores(rng, n)
    for thr in (0.3, 0.5):
        a, c = m(cu(b), cu(s), thr), scripted(cu(b), cu(s), thr)
        assert torch.equal(a, c) and a.dtype == torch.int64
        assert np.array_equal(a.cpu().numpy(), oracle.nms_rotated(b, s, thr))


def test_nms_category_id_out_of_range_raises():
    with pytest.raises(RuntimeError):
        batched_nms(torch.rand(4, 4, device=DEV), torch.rand(4, device=DEV),
                    torch.tensor([0, 1, 70000, 2], device=DEV), 0.5)


# ======================================================================== paste_masks
def test_paste_masks_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "paste_masks.npz"))
    h, w = int(d["shape"][0]), int(d["shape"][1])
    n = d["masks"].shape[0]
    exp = np.unpackbits(d["out_bits"])[: n * h * w].reshape(n, h, w).astype(bool)
    got = paste_masks_in_image(cu(d["masks"]), cu(d["boxes"]), (h, w), 0.5)
    assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), exp)
    # threshold < 0 (soft uint8): device tensors follow the reference's DEVICE path, which samples the whole image
    # (mask_ops.py:116-119): compare with the oracle's skip_empty=False mode (pinned by paste_masks_full.npz)
    got8 = paste_masks_in_image(cu(d["masks"]), Boxes(cu(d["boxes"])), (h, w), -1)
    assert got8.dtype == torch.uint8
    assert np.array_equal(got8.cpu().numpy(), oracle.paste_masks_in_image(d["masks"], d["boxes"], (h, w), -1,
                                                                          skip_empty=False))
    # odd plane sizes exercise the 4-byte and 1-byte store paths
    for hh, ww in [(37, 41), (30, 50), (64, 64)]:
        g = paste_masks_in_image(cu(d["masks"]), cu(d["boxes"] * 0.3), (hh, ww), 0.5).cpu().numpy()
        assert np.array_equal(g, oracle.paste_masks_in_image(d["masks"], d["boxes"] * 0.3, (hh, ww), 0.5))


def test_paste_masks_device_path_golden(golden_dir):
    """Large boxes, threshold 0.1 and the soft uint8 output: bit-exact against the reference's own
    `_do_paste_mask(..., skip_empty=False)` -- the region kernel covers the half-mask-pixel margin outside the box."""
    d = np.load(os.path.join(golden_dir, "paste_masks_full.npz"))
    h, w = int(d["shape"][0]), int(d["shape"][1])
    n = d["masks"].shape[0]
    got8 = paste_masks_in_image(cu(d["masks"]), cu(d["boxes"]), (h, w), -1)
    assert np.array_equal(got8.cpu().numpy(), d["out_u8"])
    for thr, key in ((0.1, "out_thr01"), (0.5, "out_thr05")):
        exp = np.unpackbits(d[key])[: n * h * w].reshape(n, h, w).astype(bool)
        assert np.array_equal(paste_masks_in_image(cu(d["masks"]), cu(d["boxes"]), (h, w), thr).cpu().numpy(), exp)


def test_paste_masks_full_size():
    """SURVEY 8d paste micro: N=100, 28x28 -> 800x1333, bit-exact vs oracle; + properties."""
    torch.manual_seed(42)
    n, H, W = 100, 800, 1333
    masks = torch.rand(n, 28, 28)
    xy = torch.rand(n, 2) * torch.tensor([W * 0.8, H * 0.8])
    wh = torch.rand(n, 2) * torch.tensor([W * 0.4, H * 0.4]) + 4
    boxes = torch.cat([xy, xy + wh], 1)
    got = paste_masks_in_image(masks.to(DEV), boxes.to(DEV), (H, W), 0.5)
    exp = oracle.paste_masks_in_image(masks.numpy(), boxes.numpy(), (H, W), 0.5)
    assert np.array_equal(got.cpu().numpy(), exp)
    # an all-ones mask pastes exactly the pixels whose centre maps inside the mask grid hull
    ones = paste_masks_in_image(torch.ones(n, 28, 28, device=DEV), boxes.to(DEV), (H, W), 0.5)
    area = ones.flatten(1).sum(1).float().cpu()
    box_area = (boxes[:, 2].clamp(max=W) - boxes[:, 0]) * (boxes[:, 3].clamp(max=H) - boxes[:, 1])
    assert torch.all((area - box_area).abs() <= 0.08 * box_area + 64)


# ======================================================================== deformable conv
DCN_GOLDEN = np.array([[30, 41.25, 48.75, 45, 28.75], [62.25, 81, 90, 80.25, 50.25],
                       [99.75, 126, 135, 117.75, 72.75], [105, 131.25, 138.75, 120, 73.75],
                       [71.75, 89.25, 93.75, 80.75, 49.5]], np.float32)


def test_deform_conv_golden_gpu():
    # /root/reference/tests/layers/test_deformable.py:16-58
    x = torch.arange(25, dtype=torch.float32, device=DEV).reshape(1, 1, 5, 5)
    off = torch.full((1, 18, 5, 5), 0.5, device=DEV)
    d = DeformConv(1, 1, kernel_size=3, padding=1).to(DEV)
    d.weight = torch.nn.Parameter(torch.ones_like(d.weight))
    assert np.allclose(d(x, off).detach().cpu().numpy().flatten(), DCN_GOLDEN.flatten())
    m = ModulatedDeformConv(1, 1, 3, padding=1, bias=False).to(DEV)
    m.weight = d.weight
    out = m(x, off, torch.full((1, 9, 5, 5), 0.5, device=DEV))
    assert np.allclose(out.detach().cpu().numpy().flatten(), DCN_GOLDEN.flatten() * 0.5)


def test_deform_conv_small_input_and_errors():
    # test_deformable.py:112-155
    d = DeformConv(3, 4, kernel_size=5, padding=1).to(DEV)
    with pytest.raises((RuntimeError, ValueError)):
        d(torch.rand(1, 3, 2, 2, device=DEV), torch.zeros(1, 50, 0, 0, device=DEV))
    x = torch.rand(2, 4, 8, 8, device=DEV)
    with pytest.raises(RuntimeError):
        DeformConv(4, 4, 3, padding=1).to(DEV)(x, torch.zeros(2, 17, 8, 8, device=DEV))
    with pytest.raises(RuntimeError):
        ModulatedDeformConv(4, 4, 3, padding=1).to(DEV)(x, torch.zeros(2, 18, 8, 8, device=DEV),
                                                       torch.zeros(2, 8, 8, 8, device=DEV))


@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("B,C,Co,H,W,groups,dg,stride,pad,dil", [
    (2, 4, 6, 7, 9, 1, 1, 1, 1, 1),
    (2, 4, 6, 7, 9, 2, 2, 1, 1, 1),
    (1, 6, 4, 9, 8, 2, 1, 2, 1, 1),
    (2, 64, 96, 12, 15, 1, 1, 1, 1, 1),
    (1, 128, 64, 10, 11, 1, 2, 1, 2, 2),
    (3, 72, 40, 6, 7, 1, 1, 1, 0, 1),
])
def test_deform_conv_fwd_bwd_fp32(modulated, B, C, Co, H, W, groups, dg, stride, pad, dil):
    rng = np.random.default_rng(1000 + C + Co)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    off = (rng.standard_normal((B, dg * 18, Ho, Wo)) * 1.5).astype(np.float32)
    msk = (1 / (1 + np.exp(-rng.standard_normal((B, dg * 9, Ho, Wo))))).astype(np.float32) if modulated else None
    w = (rng.standard_normal((Co, C // groups, 3, 3)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(Co).astype(np.float32) if modulated else None
    kw = dict(stride=stride, padding=pad, dilation=dil, groups=groups, deformable_groups=dg)
    if modulated:
        mod = ModulatedDeformConv(C, Co, 3, bias=True, **kw).to(DEV)
        mod.bias.data = cu(bias)
    else:
        mod = DeformConv(C, Co, 3, **kw).to(DEV)
    mod.weight.data = cu(w)
    xt, ot = cu(x).requires_grad_(True), cu(off).requires_grad_(True)
    mt = cu(msk).requires_grad_(True) if modulated else None
    y = mod(xt, ot, mt) if modulated else mod(xt, ot)
    exp = oracle.deform_conv_forward(x, off, w, mask=msk, bias=bias, **kw)
    assert_close_fp32(y.detach().cpu().numpy(), exp, "parity:553")
    go = rng.standard_normal(exp.shape).astype(np.float32)
    y.backward(cu(go))
    g = oracle.deform_conv_backward(x, off, w, go, mask=msk, with_bias=modulated, **kw)
    assert_close_fp32(xt.grad.cpu().numpy(), g["grad_input"], "parity:557")
    assert_close_fp32(ot.grad.cpu().numpy(), g["grad_offset"], "parity:558")
    assert_close_fp32(mod.weight.grad.cpu().numpy(), g["grad_weight"], "parity:559")
    if modulated:
        assert_close_fp32(mt.grad.cpu().numpy(), g["grad_mask"], "parity:561")
        assert_close_fp32(mod.bias.grad.cpu().numpy(), g["grad_bias"], "parity:562")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_deform_conv_16bit(dtype):
    """16-bit I/O (MFMA bf16/f16 path) vs the oracle on the same rounded inputs."""
    torch.manual_seed(3)
    B, C, Co, H, W = 2, 64, 64, 14, 17
    q = lambda t: t.to(dtype)
    x, off = q(torch.randn(B, C, H, W)), q(torch.randn(B, 18, H, W) * 1.5)
    msk, w, bias = q(torch.sigmoid(torch.randn(B, 9, H, W))), q(torch.randn(Co, C, 3, 3) * 0.05), q(torch.randn(Co))
    mod = ModulatedDeformConv(C, Co, 3, padding=1, bias=True).to(DEV).to(dtype)
    mod.weight.data, mod.bias.data = w.to(DEV), bias.to(DEV)
    xt, ot, mt = [t.to(DEV).requires_grad_(True) for t in (x, off, msk)]
    y = mod(xt, ot, mt)
    assert y.dtype == dtype
    f = lambda t: t.float().numpy()
    exp = oracle.deform_conv_forward(f(x), f(off), f(w), mask=f(msk), bias=f(bias), padding=1)
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert rel_err(y.float().detach().cpu().numpy(), exp) < tol
    go = q(torch.randn(B, Co, H, W))
    y.backward(go.to(DEV))
    g = oracle.deform_conv_backward(f(x), f(off), f(w), f(go), mask=f(msk), with_bias=True, padding=1)
    assert rel_err(xt.grad.float().cpu().numpy(), g["grad_input"]) < tol
    assert rel_err(ot.grad.float().cpu().numpy(), g["grad_offset"]) < tol
    assert rel_err(mt.grad.float().cpu().numpy(), g["grad_mask"]) < tol
    assert rel_err(mod.weight.grad.float().cpu().numpy(), g["grad_weight"]) < tol
    assert rel_err(mod.bias.grad.float().cpu().numpy(), g["grad_bias"]) < tol


def test_deform_conv_zero_offset_equals_conv2d_full_size():
    """BASELINE config 5 res4 shape (2,256,50,84), bf16 + fp32: zero offsets and unit mask reduce
    DCN to a plain convolution (size-independent property checked against torch conv2d), and the
    op is linear in the mask."""
    torch.manual_seed(5)
    B, C, H, W = 2, 256, 50, 84
    x = torch.randn(B, C, H, W, device=DEV)
    w = torch.randn(C, C, 3, 3, device=DEV) * 0.02
    off = torch.zeros(B, 18, H, W, device=DEV)
    one = torch.ones(B, 9, H, W, device=DEV)
    y = layers.modulated_deform_conv(x, off, one, w, None, 1, 1, 1, 1, 1)
    ref = torch.nn.functional.conv2d(x, w, padding=1)
    assert_close_fp32(y.cpu().numpy(), ref.cpu().numpy(), "parity:604", floor=SUM_FLOOR)
    y1 = layers.deform_conv(x, off, w, 1, 1, 1, 1, 1)
    assert_close_fp32(y1.cpu().numpy(), ref.cpu().numpy(), "parity:606", floor=SUM_FLOOR)
    yh = layers.modulated_deform_conv(x, off, 0.5 * one, w, None, 1, 1, 1, 1, 1)
    assert_close_fp32(yh.cpu().numpy(), 0.5 * ref.cpu().numpy(), "parity:608", floor=SUM_FLOOR)
    xb, wb = x.bfloat16(), w.bfloat16()
    yb = layers.modulated_deform_conv(xb, off.bfloat16(), one.bfloat16(), wb, None, 1, 1, 1, 1, 1)
    refb = torch.nn.functional.conv2d(xb.float(), wb.float(), padding=1)
    assert rel_err(yb.float().cpu().numpy(), refb.cpu().numpy()) < 2e-2


# ======================================================================== registered ops
def test_registered_ops_scriptable():
    """torch.ops.detectron2.* stay TorchScript-callable (test_nms_rotated.py:153-168,
    roi_align_rotated.py:88-91)."""
    @torch.jit.script
    def f(b: torch.Tensor, s: torch.Tensor):
        return torch.ops.detectron2.nms_rotated(b, s, 0.5)

    rng = np.random.default_rng(9)
    b = np.stack([rng.uniform(0, 50, 40), rng.uniform(0, 50, 40), rng.uniform(2, 30, 40), rng.uniform(2, 30, 40),
                  rng.uniform(-90, 90, 40)], 1).astype(np.float32)
    s = _distinct_scores(rng, 40)
    assert np.array_equal(f(cu(b), cu(s)).cpu().numpy(), oracle.nms_rotated(b, s, 0.5))
    iou = torch.ops.detectron2.box_iou_rotated(cu(b), cu(b[:7]))
    assert np.array_equal(iou.cpu().numpy(), oracle.box_iou_rotated(b, b[:7]))


def test_roi_align_rotated_negative_size_raises():
    """ROIAlignRotated_cpu.cpp:236-238: AT_ASSERTM(roi_width >= 0 && roi_height >= 0) -> RuntimeError."""
    from detectron2_amd.layers import ROIAlignRotated

    x = torch.randn(1, 4, 16, 16, device=DEV)
    good = torch.tensor([[0, 8.0, 8.0, 6.0, 4.0, 30.0]], device=DEV)
    assert ROIAlignRotated((3, 3), 1.0, 2)(x, good).shape == (1, 4, 3, 3)
    bad = torch.tensor([[0, 8.0, 8.0, -6.0, 4.0, 30.0]], device=DEV)
    with pytest.raises(RuntimeError, match="non-negative size"):
        ROIAlignRotated((3, 3), 1.0, 2)(x, bad)


def test_batched_nms_and_paste_masks_are_scriptable_with_identical_results():
    """The reference's own contracts: `torch.jit.script(batched_nms)` == eager for N = 2000, 50 classes, thresholds
    .2 / .5 / .8, inputs not mutated (tests/layers/test_nms.py:16-29); `torch.jit.script(paste_masks_in_image)`
    bit-identical to eager, 10 masks 28x28 -> 150x150 (tests/layers/test_mask_ops.py:156-165)."""
    from detectron2_amd.layers.mask_ops import _paste_masks_tensor_shape, pad_masks, scale_boxes

    torch.manual_seed(0)
    n = 2000
    boxes = torch.rand(n, 4, device=DEV) * 100
    boxes[:, 2:] += boxes[:, :2]
    scores = torch.rand(n, device=DEV)
    idxs = torch.randint(0, 50, (n,), device=DEV)
    scripted = torch.jit.script(batched_nms)
    for thr in (0.2, 0.5, 0.8):
        backup = boxes.clone()
        a, b = batched_nms(boxes, scores, idxs, thr), scripted(boxes, scores, idxs, thr)
        assert torch.equal(boxes, backup) and torch.equal(a, b) and a.dtype == torch.int64
        assert np.array_equal(a.cpu().numpy(), oracle.batched_nms(boxes.cpu().numpy(), scores.cpu().numpy(),
                                                                 idxs.cpu().numpy(), thr))
    masks = torch.rand(10, 28, 28, device=DEV)
    pb = torch.rand(10, 4, device=DEV) * 60
    pb[:, 2:] = pb[:, :2] + 20 + pb[:, 2:]
    sp = torch.jit.script(paste_masks_in_image)
    for thr in (0.5, -1.0):
        e, s2 = paste_masks_in_image(masks, pb, (150, 150), thr), sp(masks, pb, (150, 150), thr)
        assert tuple(e.shape) == (10, 150, 150) and e.dtype == s2.dtype and torch.equal(e, s2)
    assert sp(masks[:0], pb[:0], (150, 150), 0.5).shape == (0, 150, 150)
    t = _paste_masks_tensor_shape(masks, pb, (torch.tensor(150), torch.tensor(150)), 0.5)
    assert torch.equal(t, paste_masks_in_image(masks, pb, (150, 150), 0.5))
    # helpers of the module (mask_ops.py:219-262)
    pm, scale = pad_masks(masks, 1)
    assert pm.shape == (10, 30, 30) and scale == 30 / 28 and torch.equal(pm[:, 1:-1, 1:-1], masks) and pm[:, 0].sum() == 0
    sb = scale_boxes(pb, scale)
    assert torch.allclose((sb[:, 2:] - sb[:, :2]), (pb[:, 2:] - pb[:, :2]) * scale, rtol=1e-5)
    assert torch.allclose((sb[:, 2:] + sb[:, :2]), (pb[:, 2:] + pb[:, :2]), rtol=1e-5)


def test_fp16_rois_are_not_rounded_to_the_feature_dtype_and_what_that_changes():
    """The reference casts the ROIs to the input dtype before the op (layers/roi_align.py:60): with fp16 features an ROI
    coordinate of 1,200 px is then a multiple of 1 px, one of 300 px a multiple of 0.25 px.  This package keeps the ROIs
    in fp32 (SURVEY 7 sanctions it for bf16 and the same policy is applied to fp16).  Stated here by how much
    that differs on the same inputs: the op equals the oracle on the UNROUNDED boxes to the fp16 output rounding, and
    the oracle on fp16-rounded boxes -- what the reference computes -- is up to several 1e-2 of the feature range away
    for large image coordinates (and identical for boxes that are exact in fp16)."""
    rng = np.random.default_rng(77)
    N, C, H, W = 1, 8, 200, 336                      # p2 of an 800 x 1344 image
    x = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float16)
    rois = random_rois(rng, 64, N, W, H, 0.25, 16.0)  # image coordinates up to 1,344: fp16 ulp 0.5 .. 1 px
    xt = torch.from_numpy(x).to(DEV)
    y = ROIAlign((7, 7), 0.25, 0, True)(xt, torch.from_numpy(rois).to(DEV)).float().cpu().numpy()
    exact = oracle.roi_align_forward(x.astype(np.float32), rois, (7, 7), 0.25, 0, True)
    assert np.abs(y - exact).max() <= 2.0 ** -10 * np.abs(exact).max() + 1e-4     # fp16 output rounding only
    rounded = rois.copy()
    rounded[:, 1:] = rois[:, 1:].astype(np.float16).astype(np.float32)
    ref_like = oracle.roi_align_forward(x.astype(np.float32), rounded, (7, 7), 0.25, 0, True)
    gap = float(np.abs(ref_like - exact).max())
    assert 1e-3 < gap < 0.5, gap      # a real, bounded difference: the reference's own result moves by this much
    exact_in_fp16 = np.round(rois * 4) / 4
    exact_in_fp16[:, 1:] = np.clip(exact_in_fp16[:, 1:], 0, 500)   # < 512: quarter pixels are exact in fp16
    y2 = ROIAlign((7, 7), 0.25, 0, True)(xt, torch.from_numpy(exact_in_fp16.astype(np.float32)).to(DEV)).float().cpu().numpy()
    same = oracle.roi_align_forward(x.astype(np.float32), exact_in_fp16.astype(np.float16).astype(np.float32), (7, 7), 0.25, 0, True)
    assert np.abs(y2 - same).max() <= 2.0 ** -10 * np.abs(same).max() + 1e-4


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.7])
def test_nms_rotated_sub_pixel_boxes(thr):
    """boxes of 0.01-0.05 px -- the reference's clip tolerance (EPS 1e-5 ABSOLUTE) is up to 10 % of their sides,
    so their IoU can exceed the area ratio: the mask kernel's area-ratio shortcut must not apply to them (it is limited to
    sides >= 1 px).  Kept sets equal the oracle's full clip, bit for bit."""
    rng = np.random.default_rng(int(thr * 1000))
    boxes, scores = [], []
    for g in range(200):
        cx, cy = rng.uniform(10, 11, 2)
        w, h = rng.uniform(0.01, 0.05, 2)
        ang = rng.uniform(-90, 90)
        boxes.append([cx, cy, w, h, ang]); scores.append(1.0 - 1e-3 * g)
        boxes.append([cx + rng.uniform(-0.002, 0.002), cy + rng.uniform(-0.002, 0.002), w * rng.uniform(0.6, 1.9),
                      h * rng.uniform(0.6, 1.9), ang + rng.uniform(-3, 3)])
        scores.append(0.5 - 1e-3 * g)
    b = np.asarray(boxes, np.float32)
    s = np.asarray(scores, np.float32)
    want = oracle.nms_rotated(b, s, thr)
    got = nms_rotated(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), thr).cpu().numpy()
    assert np.array_equal(got, want), (len(got), len(want))
    idx = np.zeros(len(b), np.int64)
    got_b = layers.batched_nms_rotated(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), torch.from_numpy(idx).to(DEV), thr)
    assert np.array_equal(got_b.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_reference_roi_rounding_switch(dtype, monkeypatch):
    """D2AMD_REFERENCE_ROI_ROUNDING=1 / _C.set_reference_roi_rounding (missing 4): the ROIs are rounded to the feature dtype before pooling,
    as layers/roi_align.py:60 does -- ROIAlign and the fused ROIPooler then equal the oracle on the ROUNDED boxes (what
    the reference computes) to the 16-bit output rounding, and differ from the default (unrounded) result."""
    from detectron2_amd.modeling import ROIPooler
    from detectron2_amd.structures import Boxes

    rng = np.random.default_rng(78)
    N, C, H, W = 1, 8, 200, 336
    x = torch.from_numpy(rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)).to(dtype)
    rois = random_rois(rng, 64, N, W, H, 0.25, 16.0)
    xt = x.to(DEV)
    rt = torch.from_numpy(rois).to(DEV)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    plain = ROIAlign((7, 7), 0.25, 0, True)(xt, rt).float().cpu().numpy()
    from detectron2_amd import _C as _lib

    monkeypatch.setattr(_lib, "_REFERENCE_ROI_ROUNDING", True)
    strict = ROIAlign((7, 7), 0.25, 0, True)(xt, rt).float().cpu().numpy()
    rounded = torch.from_numpy(rois).to(dtype).float().numpy()
    want = oracle.roi_align_forward(x.float().numpy(), rounded, (7, 7), 0.25, 0, True)
    assert np.abs(strict - want).max() <= ulp * np.abs(want).max() + 1e-4
    assert np.abs(strict - plain).max() > 1e-3  # (coordinates up to 1,344 px: the rounding moves the samples)
    # the fused pooler (one level here: canonical size chosen so that every box maps to p2)
    pooler = ROIPooler((7, 7), (0.25,), 0, "ROIAlignV2")
    boxes = [Boxes(rt[:, 1:].clone())]
    got = pooler([xt.contiguous(memory_format=torch.channels_last)], boxes).float().cpu().numpy()
    assert np.abs(got - want).max() <= ulp * np.abs(want).max() + 1e-4
    monkeypatch.setattr(_lib, "_REFERENCE_ROI_ROUNDING", False)
    got0 = pooler([xt.contiguous(memory_format=torch.channels_last)], boxes).float().cpu().numpy()
    assert np.abs(got0 - plain).max() <= ulp * np.abs(plain).max() + 1e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_reference_roi_rounding_keeps_the_levels_of_the_unrounded_boxes(dtype, monkeypatch):
    """ROIPooler.forward (poolers.py:240-262) assigns levels from the fp32 boxes and only each level's
    ROIAlign casts its ROIs to the feature dtype (roi_align.py:60).  Boxes whose sqrt(area) sits within the 16-bit
    rounding of a level threshold (112 / 224 / 448 px) would change level if they were rounded FIRST: in strict mode the
    fused pooler must pool every box on the level of its UNROUNDED size, with ROUNDED coordinates -- forward against the
    oracle per level, backward (16-bit tile gather, strict mode) against the oracle's scatter of the same rows."""
    from detectron2_amd import _C as _lib
    from detectron2_amd.modeling import ROIPooler
    from detectron2_amd.structures import Boxes
    from test_tile_gather_math import assign_levels_restated

    rng = np.random.default_rng(5)
    N, C = 1, 32
    hw = [(200, 336), (100, 168), (50, 84), (25, 42)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    feats = [rng.uniform(-1, 1, (N, C, h, w)).astype(np.float32) for h, w in hw]
    boxes = []
    for thr in (112.0, 224.0, 448.0):  # square-ish boxes a hair below / above each threshold, far from the origin
        for _ in range(24):
            s = thr * (1.0 + rng.uniform(-2e-3, 2e-3))
            a = rng.uniform(0.8, 1.25)
            w_, h_ = s * np.sqrt(a), s / np.sqrt(a)
            x1, y1 = rng.uniform(300, 1300 - w_), rng.uniform(100, 790 - h_)
            boxes.append([x1, y1, x1 + w_, y1 + h_])
    boxes = np.asarray(boxes, np.float32)
    lv = assign_levels_restated(boxes, 2, 5, 224, 4)
    lv_rounded = assign_levels_restated(torch.from_numpy(boxes).to(dtype).float().numpy(), 2, 5, 224, 4)
    assert (lv != lv_rounded).sum() >= 3, "the case must hold boxes whose level the rounding would change"
    rounded = torch.from_numpy(boxes).to(dtype).float().numpy()
    xs = [torch.from_numpy(f).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
    monkeypatch.setattr(_lib, "_REFERENCE_ROI_ROUNDING", True)
    pooler = ROIPooler((7, 7), scales, 0, "ROIAlignV2")
    y = pooler(xs, [Boxes(torch.from_numpy(boxes).to(DEV))])
    got = y.detach().float().cpu().numpy()
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    g = rng.standard_normal(got.shape).astype(np.float32)
    y.backward(torch.from_numpy(g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last))
    g16 = torch.from_numpy(g).to(dtype).float().numpy()
    for l in range(4):
        rows = np.nonzero(lv == l)[0]
        rois = np.concatenate([np.zeros((len(rows), 1), np.float32), rounded[rows]], 1)
        f16 = torch.from_numpy(feats[l]).to(dtype).float().numpy()
        want = oracle.roi_align_forward(f16, rois, (7, 7), scales[l], 0, True)
        assert np.abs(got[rows] - want).max() <= ulp * np.abs(want).max() + 1e-4, l
        gin = oracle.roi_align_backward(np.ascontiguousarray(g16[rows]), rois, f16.shape, scales[l], 0, True)
        d = np.abs(xs[l].grad.float().cpu().numpy() - gin)
        assert d.max() <= 2 * ulp * np.abs(gin).max() + 1e-4, (l, float(d.max()))


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.7, 0.9])
def test_nms_rotated_pairs_at_the_area_ratio_bound(thr):
    """The rotated mask kernel skips the polygon clip for pairs whose area ratio is below 0.99 x threshold (IoU <= min / max
    area) and for pairs the centre-distance test rejects.  Concentric boxes of equal angle have IoU == area ratio: a ladder
    of ratios from 0.97 x to 1.01 x the threshold (and the same ladder slightly rotated / shifted) must give the kept set
    of the oracle's full clip (the restatement of nms_rotated_cpu.cpp), bit for bit."""
    rng = np.random.default_rng(int(thr * 100))
    boxes, scores = [], []
    for g in range(120):
        cx, cy = rng.uniform(50, 950, 2)
        w, h = np.exp(rng.uniform(np.log(10), np.log(200), 2))
        ang = rng.uniform(-90, 90)
        boxes.append([cx, cy, w, h, ang]); scores.append(1.0 - 1e-3 * g)
        ratio = thr * rng.uniform(0.97, 1.01)
        s = np.sqrt(ratio)
        jitter = rng.choice([0.0, 0.0, 0.3, 1.0])
        boxes.append([cx + jitter * rng.uniform(-1, 1), cy + jitter * rng.uniform(-1, 1), w * s, h * s,
                      ang + jitter * rng.uniform(-2, 2)])
        scores.append(0.5 - 1e-3 * g)
    b = np.asarray(boxes, np.float32)
    s = np.asarray(scores, np.float32)
    want = oracle.nms_rotated(b, s, thr)
    got = nms_rotated(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), thr).cpu().numpy()
    assert np.array_equal(got, want)
    assert 120 < len(want) < 240  # some of the smaller boxes are suppressed, some are not


@pytest.mark.parametrize("thr", [0.1, 0.3, 0.5, 0.7, 0.9])
def test_nms_rotated_pairs_at_the_projection_bound(thr):
    """r06: the rotated mask tile also skips the clip when the PROJECTION bound of the intersection (overlap of the two boxes'
    projections on the axes of either box) gives an IoU more than 1 % below the threshold.  For two boxes of equal angle
    shifted along an axis the bound IS the intersection: ladders of shifts whose IoU runs from 0.97 x to 1.01 x the
    threshold -- equal angle, slightly different angles, crossing elongated boxes -- and RRPN-like clusters (shared centres,
    angles every 30 degrees, aspect ratios up to 1:8) must give the kept set of the oracle's full clip."""
    rng = np.random.default_rng(1000 + int(thr * 100))
    boxes, scores = [], []
    for g in range(60):  # shift ladders: IoU = (w - d) / (w + d) for equal boxes shifted by d along the width axis
        cx, cy = rng.uniform(100, 900, 2)
        w, h = np.exp(rng.uniform(np.log(8), np.log(300))), np.exp(rng.uniform(np.log(4), np.log(60)))
        ang = rng.uniform(-180, 180)
        iou = thr * rng.uniform(0.97, 1.01)
        d = w * (1 - iou) / (1 + iou)
        t = np.deg2rad(ang)
        ux, uy = np.cos(t), -np.sin(t)  # the width axis of a box (rot_vertices)
        boxes.append([cx, cy, w, h, ang]); scores.append(1.0 - 1e-3 * g)
        dang = rng.choice([0.0, 0.0, 0.5, 3.0]) * rng.uniform(-1, 1)
        boxes.append([cx + d * ux, cy + d * uy, w, h, ang + dang]); scores.append(0.6 - 1e-3 * g)
    for g in range(40):  # crossing boxes: a long thin one over a wider one, the angle between them 60-90 degrees
        cx, cy = rng.uniform(100, 900, 2)
        w1, h1 = rng.uniform(60, 200), rng.uniform(20, 60)
        ang = rng.uniform(-90, 90)
        boxes.append([cx, cy, w1, h1, ang]); scores.append(0.5 - 1e-3 * g)
        boxes.append([cx + rng.uniform(-3, 3), cy + rng.uniform(-3, 3), h1 * rng.uniform(0.8, 1.6), w1 * rng.uniform(0.3, 1.2),
                      ang + rng.uniform(60, 90)])
        scores.append(0.4 - 1e-3 * g)
    for g in range(12):  # RRPN-like clusters
        cx, cy = rng.uniform(200, 800, 2)
        size = rng.uniform(32, 128)
        for a in range(-90, 90, 30):
            for ar in (0.125, 0.5, 1.0, 2.0):
                boxes.append([cx + rng.uniform(-8, 8), cy + rng.uniform(-8, 8), size * np.sqrt(ar) * rng.uniform(0.8, 1.25),
                              size / np.sqrt(ar) * rng.uniform(0.8, 1.25), a + rng.uniform(-10, 10)])
                scores.append(rng.uniform(0.0, 0.3))
    b = np.asarray(boxes, np.float32)
    s = np.asarray(scores, np.float32)
    want = oracle.nms_rotated(b, s, thr)
    got = nms_rotated(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), thr).cpu().numpy()
    assert np.array_equal(got, want)
    assert 60 < len(want) < len(b)
