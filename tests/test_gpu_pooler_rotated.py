"""GPU parity of the fused multi-level ROIAlignRotated pooler (csrc/roi_pool_rot.hip behind
detectron2_amd.modeling.ROIPooler(pooler_type="ROIAlignRotated") on channels_last features) against the C oracle
(oracle.roi_align_rotated_forward / _backward: the restatement of csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp, pinned
to the compiled reference in tests/test_oracle_golden.py) applied level by level with the reference's level assignment
(modeling/poolers.py:51-59 restated in numpy).  fp32: per-element 1e-4 |ref| + ROI_FLOOR max|ref|; 16-bit: 2 ulp of the
dtype on the fp32 result of the same (rounded) inputs + the same floor."""
import numpy as np
import pytest
import torch

import oracle
from conftest import ROI_FLOOR, assert_close_fp32
from detectron2_amd.modeling import ROIPooler

pytestmark = pytest.mark.gpu
DEV = "cuda"
STRIDES = (4, 8, 16, 32)


class RotatedBoxes:
    """(cx, cy, w, h, angle) rows with the members ROIPooler uses of the reference's structures.RotatedBoxes."""

    def __init__(self, tensor):
        self.tensor = tensor

    def area(self):
        return self.tensor[:, 2] * self.tensor[:, 3]

    def __len__(self):
        return self.tensor.shape[0]


def levels_of(boxes, min_level=2, max_level=5, canonical_size=224, canonical_level=4):
    sizes = np.sqrt((boxes[:, 2] * boxes[:, 3]).astype(np.float32))
    lv = np.floor(np.float32(canonical_level) + np.log2(sizes / np.float32(canonical_size) + np.float32(1e-8)))
    return (np.clip(lv, min_level, max_level) - min_level).astype(np.int64)


def make(seed, n_img=2, c=32, base=(96, 128), n_box=40, big=False):
    rng = np.random.default_rng(seed)
    feats = [rng.standard_normal((n_img, c, base[0] * 4 // s, base[1] * 4 // s)).astype(np.float32) for s in STRIDES]
    boxes = []
    for _ in range(n_img):
        ctr = rng.uniform([0, 0], [base[1] * 4, base[0] * 4], (n_box, 2))
        lo, hi = (np.log(40), np.log(700)) if big else (np.log(6), np.log(600))
        wh = np.exp(rng.uniform(lo, hi, (n_box, 2)))
        ang = rng.uniform(-180, 180, (n_box, 1))
        b = np.concatenate([ctr, wh, ang], 1).astype(np.float32)
        b[0, 2:4] = [0.5, 0.7]      # a tiny box
        b[1, 4] = 0.0               # axis-aligned
        b[2, 4] = 90.0
        b[3, :2] = [-30.0, -20.0]   # centre outside the image
        boxes.append(b)
    return feats, boxes


def oracle_pool(feats, boxes, out, sr):
    rois = np.concatenate([np.concatenate([np.full((len(b), 1), i, np.float32), b], 1) for i, b in enumerate(boxes)])
    lv = levels_of(rois[:, 1:])
    res = np.zeros((len(rois), feats[0].shape[1], out, out), np.float32)
    for l, s in enumerate(STRIDES):
        idx = np.nonzero(lv == l)[0]
        if len(idx):
            res[idx] = oracle.roi_align_rotated_forward(feats[l], rois[idx], (out, out), 1.0 / s, sr)
    return res, rois, lv


def run_fused(feats, boxes, out, sr, dtype, grad=None):
    x = [torch.from_numpy(f).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(grad is not None)
         for f in feats]
    pooler = ROIPooler(out, [1.0 / s for s in STRIDES], sr, "ROIAlignRotated")
    y = pooler(x, [RotatedBoxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    gx = None
    if grad is not None:
        y.backward(torch.from_numpy(grad).to(DEV).to(dtype))
        gx = [t.grad.float().cpu().numpy() for t in x]
    return y.float().detach().cpu().numpy(), gx


@pytest.mark.parametrize("out,sr,big", [(7, 2, False), (7, 0, False), (14, 2, False), (7, 0, True), (3, 3, False)])
def test_forward_fp32_vs_oracle(out, sr, big):
    feats, boxes = make(1 + out + sr, big=big)
    want, _, lv = oracle_pool(feats, boxes, out, sr)
    assert len(set(lv.tolist())) >= 3  # the boxes spread over the levels
    got, _ = run_fused(feats, boxes, out, sr, torch.float32)
    assert_close_fp32(got, want, f"rot_pooler_fwd_{out}_{sr}_{big}", floor=ROI_FLOOR)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_forward_16bit_vs_oracle(dtype):
    feats, boxes = make(5)
    feats = [torch.from_numpy(f).to(dtype).float().numpy() for f in feats]  # the values the kernel reads
    want, _, _ = oracle_pool(feats, boxes, 7, 2)
    got, _ = run_fused(feats, boxes, 7, 2, dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    bound = 2 * ulp * np.abs(want) + ROI_FLOOR * np.abs(want).max() + 1e-4 * np.abs(want)
    assert (np.abs(got - want) <= bound).all(), float((np.abs(got - want) / bound).max())


@pytest.mark.parametrize("out,sr,big", [(7, 2, False), (7, 0, True)])
def test_backward_fp32_vs_oracle(out, sr, big):
    feats, boxes = make(11 + out, c=16, n_box=24, big=big)
    rng = np.random.default_rng(7)
    _, rois, lv = oracle_pool(feats, boxes, out, sr)
    gy = rng.standard_normal((len(rois), 16, out, out)).astype(np.float32)
    _, gx = run_fused(feats, boxes, out, sr, torch.float32, grad=gy)
    for l, s in enumerate(STRIDES):
        idx = np.nonzero(lv == l)[0]
        want = np.zeros_like(feats[l])
        if len(idx):
            want = oracle.roi_align_rotated_backward(gy[idx], rois[idx], feats[l].shape, 1.0 / s, sr)
        # fp32 sums of up to hundreds of scattered terms in an arbitrary order on both sides
        assert_close_fp32(gx[l], want, f"rot_pooler_bwd_{out}_{sr}_{big}_p{l + 2}", floor=4 * ROI_FLOOR)


def test_backward_bf16_matches_the_fp32_path():
    feats, boxes = make(21, c=32, n_box=30)
    rng = np.random.default_rng(9)
    gy = rng.standard_normal((60, 32, 7, 7)).astype(np.float32)
    gy = torch.from_numpy(gy).to(torch.bfloat16).float().numpy()
    _, g32 = run_fused(feats, boxes, 7, 2, torch.float32, grad=gy)
    _, g16 = run_fused(feats, boxes, 7, 2, torch.bfloat16, grad=gy)
    for a, b in zip(g16, g32):  # one rounding of the fp32 sum
        assert (np.abs(a - b) <= 2.0 ** -8 * np.abs(b) + 1e-5 * np.abs(b).max()).all()


def test_negative_size_raises_and_empty_lists_work():
    feats, boxes = make(3, n_box=6)
    boxes[1][2, 2:4] = [-40.0, -30.0]  # (area > 0: it gets a level, and ROIAlignRotated_cpu.cpp:236 asserts)
    with pytest.raises(RuntimeError):
        run_fused(feats, boxes, 7, 2, torch.float32)
    feats, boxes = make(3, n_box=6)
    boxes[1][2, 2] = -4.0  # one negative side: sqrt(area) is NaN, the box matches no level, its rows stay zero (poolers.py:247-263)
    got, _ = run_fused(feats, boxes, 7, 2, torch.float32)
    assert not got[6 + 2].any() and got[6 + 3].any()
    feats, boxes = make(3, n_box=6)
    boxes[0] = boxes[0][:0]
    got, _ = run_fused(feats, boxes, 7, 2, torch.float32)
    want, _, _ = oracle_pool(feats, boxes, 7, 2)
    assert got.shape == want.shape == (6, 32, 7, 7)
    assert_close_fp32(got, want, "rot_pooler_one_empty_image", floor=ROI_FLOOR)


# ---- the backward as a deterministic gather (r05) ---------------------------------------------------------------------
def _dense(seed, n_box, size, c=16, jitter=6.0):
    """n_box boxes per image of about `size` px around ONE spot: long per-pixel lists on one level."""
    rng = np.random.default_rng(seed)
    feats = [rng.standard_normal((2, c, 96 * 4 // s, 128 * 4 // s)).astype(np.float32) for s in STRIDES]
    boxes = []
    for _ in range(2):
        ctr = np.array([[250.0, 190.0]]) + rng.uniform(-jitter, jitter, (n_box, 2))
        wh = size * np.exp(rng.uniform(-0.1, 0.1, (n_box, 2)))
        ang = rng.uniform(-180, 180, (n_box, 1))
        boxes.append(np.concatenate([ctr, wh, ang], 1).astype(np.float32))
    return feats, boxes


def _bwd_vs_oracle(feats, boxes, out, sr, tag, dtype=torch.float32, floor=4 * ROI_FLOOR):
    c = feats[0].shape[1]
    _, rois, lv = oracle_pool(feats, boxes, out, sr)
    gy = np.random.default_rng(5).standard_normal((len(rois), c, out, out)).astype(np.float32)
    if dtype != torch.float32:
        gy = torch.from_numpy(gy).to(dtype).float().numpy()
    _, gx = run_fused(feats, boxes, out, sr, dtype, grad=gy)
    for l, s in enumerate(STRIDES):
        idx = np.nonzero(lv == l)[0]
        want = np.zeros_like(feats[l])
        if len(idx):
            want = oracle.roi_align_rotated_backward(gy[idx], rois[idx], feats[l].shape, 1.0 / s, sr)
        if dtype == torch.float32:
            assert_close_fp32(gx[l], want, f"{tag}_p{l + 2}", floor=floor)
        else:
            ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
            assert (np.abs(gx[l] - want) <= ulp * np.abs(want) + 1e-4 * np.abs(want).max()).all(), (tag, l)
    return gx


@pytest.mark.parametrize("n_box,size", [(40, 120.0), (160, 500.0), (700, 600.0)])
def test_backward_long_pixel_lists(n_box, size):
    """lists of tens (sorted in registers), hundreds (ranked against LDS) and thousands (against memory) of entries"""
    feats, boxes = _dense(31, n_box, size)
    _bwd_vs_oracle(feats, boxes, 7, 2, f"rot_pooler_bwd_dense_{n_box}", floor=16 * ROI_FLOOR)


def test_backward_is_deterministic_and_equals_the_atomic_path():
    import os

    feats, boxes = make(41, c=32, n_box=60)
    gy = np.random.default_rng(3).standard_normal((120, 32, 7, 7)).astype(np.float32)
    for dtype in (torch.float32, torch.bfloat16):
        runs = [run_fused(feats, boxes, 7, 0, dtype, grad=gy)[1] for _ in range(3)]
        for r in runs[1:]:
            for a, b in zip(runs[0], r):
                assert np.array_equal(a, b), "the gather backward differs between runs"
        os.environ["D2AMD_ROT_BWD_ATOMICS"] = "1"
        try:
            atom = run_fused(feats, boxes, 7, 0, dtype, grad=gy)[1]
        finally:
            del os.environ["D2AMD_ROT_BWD_ATOMICS"]
        tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
        for a, b in zip(runs[0], atom):
            assert (np.abs(a - b) <= tol * np.abs(b) + 1e-5 * np.abs(b).max()).all()


def test_backward_rois_whose_table_does_not_fit():
    """bins wider than ~8 px at their level (a 3,000-px box on p5: 14 x 14 samples per bin with sampling_ratio 0): flagged by
    the table pass, scattered with atomics into the fp32 image the gather adds -- beside ordinary ROIs on the same pixels."""
    feats, boxes = make(51, c=16, n_box=12)
    for b in boxes:
        b[5, :4] = [260.0, 200.0, 3000.0, 2600.0]
        b[6, :4] = [100.0, 300.0, 40.0, 2900.0]
    _bwd_vs_oracle(feats, boxes, 7, 0, "rot_pooler_bwd_flagged", floor=16 * ROI_FLOOR)
    _bwd_vs_oracle(feats, boxes, 7, 0, "rot_pooler_bwd_flagged_bf16", dtype=torch.bfloat16)


def test_backward_odd_channel_counts():
    feats, boxes = make(61, c=6, n_box=10)   # not a multiple of 4: the scalar lanes of the gather
    _bwd_vs_oracle(feats, boxes, 7, 2, "rot_pooler_bwd_c6")
    feats, boxes = make(62, c=260, n_box=6)  # two channel slabs
    _bwd_vs_oracle(feats, boxes, 3, 2, "rot_pooler_bwd_c260")


def test_backward_without_rois_writes_zeros():
    feats, boxes = make(71, c=16, n_box=5)
    boxes = [b[:0] for b in boxes]
    x = [torch.from_numpy(f).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
    pooler = ROIPooler(7, [1.0 / s for s in STRIDES], 2, "ROIAlignRotated")
    y = pooler(x, [RotatedBoxes(torch.from_numpy(b).to(DEV)) for b in boxes])
    assert y.shape == (0, 16, 7, 7)
    (y.sum() + sum(t.sum() * 0 for t in x)).backward()
    for t in x:
        assert t.grad is not None and not t.grad.any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_single_level_layer_backward_on_channels_last_is_the_gather(dtype):
    """layers.ROIAlignRotated on a channels_last input: its backward runs the fused pooler's gather with one level --
    equal to the NCHW entry's atomic scatter up to the order of the fp32 sums, and bit-identical from run to run."""
    from detectron2_amd.layers import ROIAlignRotated

    rng = np.random.default_rng(81)
    x = rng.standard_normal((2, 24, 40, 56)).astype(np.float32)
    n_roi = 50
    rois = np.concatenate([rng.integers(0, 2, (n_roi, 1)).astype(np.float32), rng.uniform(0, 220, (n_roi, 2)),
                           np.exp(rng.uniform(np.log(8), np.log(200), (n_roi, 2))), rng.uniform(-180, 180, (n_roi, 1))],
                          1).astype(np.float32)
    gy = rng.standard_normal((n_roi, 24, 7, 7)).astype(np.float32)
    layer = ROIAlignRotated((7, 7), 0.25, 0)

    def run(channels_last):
        t = torch.from_numpy(x).to(DEV).to(dtype)
        if channels_last:
            t = t.contiguous(memory_format=torch.channels_last)
        t.requires_grad_(True)
        y = layer(t, torch.from_numpy(rois).to(DEV))
        g = torch.from_numpy(gy).to(DEV).to(dtype)
        y.backward(g.contiguous(memory_format=torch.channels_last) if channels_last else g)
        return y.detach().float().cpu().numpy(), t.grad.float().cpu().numpy()

    y0, g0 = run(False)
    y1, g1 = run(True)
    y2, g2 = run(True)
    assert np.array_equal(g1, g2), "channels_last backward differs between runs"
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    assert np.allclose(y1, y0, rtol=tol, atol=tol * np.abs(y0).max())
    assert (np.abs(g1 - g0) <= tol * np.abs(g0) + 1e-5 * np.abs(g0).max() + (0 if dtype == torch.float32 else 2.0 ** -8 * np.abs(g0))).all()
    if dtype == torch.float32:
        want = oracle.roi_align_rotated_backward(gy, rois, x.shape, 0.25, 0)
        assert_close_fp32(g1, want, "rot_layer_bwd_nhwc", floor=4 * ROI_FLOOR)


def test_backward_gather_does_not_spread_a_non_finite_gradient_row():
    """the gather's lanes past a pixel's list re-read dY row 0 with weight 0 -- an Inf there (an fp16 / bf16
    AMP overflow) must stay on the pixels ROI 0 touches (0 x Inf = NaN would reach every pixel whose list is shorter
    than its wave's longest).  The reference's scatter only touches a sample's own taps (ROIAlignRotated_cuda.cu:224-323)."""
    feats, boxes = make(91, n_img=1, c=32, n_box=24)
    boxes[0][0] = [20.0, 20.0, 10.0, 8.0, 15.0]  # ROI 0: a small box in the top-left corner (level p2)
    g = np.random.default_rng(3).standard_normal((24, 32, 7, 7)).astype(np.float32)
    g[0] = np.inf
    _, gx = run_fused(feats, boxes, 7, 2, torch.bfloat16, grad=g)
    p2 = gx[0]
    assert np.isinf(p2[0, :, 2:8, 2:8]).any()          # the ROI's own pixels carry it
    assert not np.isnan(p2).any()
    assert np.isfinite(p2[0, :, 16:, :]).all() and np.isfinite(p2[0, :, :, 16:]).all()  # nobody else does
    for l in range(1, 4):
        assert np.isfinite(gx[l]).all(), l
