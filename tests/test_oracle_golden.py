"""Pins the CPU oracle (oracle/d2_oracle.c) against (a) every known answer the reference's own
tests hold for the hot path, (b) fixtures produced by running the reference (tests/golden/,
made by tests/golden/make_golden.py), and (c) the compiled reference (oracle/_ref) live when it
is present.  CPU only."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ref

from _torch_ref import dcn_torch


# ---------------------------------------------------------------- ROIAlign known answers
def test_roi_align_known_answers():
    # /root/reference/tests/layers/test_roi_align.py:14-47
    x = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    rois = np.array([[0, 1, 1, 3, 3]], np.float32)
    old = oracle.roi_align_forward(x, rois, (4, 4), 1.0, 0, False)[0, 0]
    new = oracle.roi_align_forward(x, rois, (4, 4), 1.0, 0, True)[0, 0]
    old_exp = [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]]
    new_exp = [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0],
               [12.0, 12.5, 13.0, 13.5]]
    assert np.allclose(old, old_exp)
    assert np.allclose(new, new_exp)


def test_roi_align_empty_box_and_grad():
    # test_roi_align.py:111-121
    rng = np.random.default_rng(0)
    x = rng.random((1, 1, 5, 5), dtype=np.float32)
    rois = np.array([[0, 3, 4, 5, 4]], np.float32)
    o = oracle.roi_align_forward(x, rois, (7, 7), 1.0, 0, True)
    assert o.shape == (1, 1, 7, 7) and (o == 0).all()
    g = oracle.roi_align_backward(np.ones_like(o), rois, x.shape, 1.0, 0, True)
    assert (g == 0).all()


def test_roi_align_equals_rotated_at_zero_angle():
    # /root/reference/tests/modeling/test_roi_pooler.py:14-59: ROIAlignV2 == ROIAlignRotated(0 deg)
    rng = np.random.default_rng(1)
    x = rng.random((2, 4, 10, 8), dtype=np.float32)
    b = rng.random((10, 4), dtype=np.float32) * 64
    b[:, 2:] = b[:, :2] + np.maximum(b[:, 2:], 1.0)
    bi = rng.integers(0, 2, 10).astype(np.float32)
    rois = np.concatenate([bi[:, None], b], 1)
    rrois = np.stack([bi, (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0],
                      b[:, 3] - b[:, 1], np.zeros(10, np.float32)], 1)
    for sr in (0, 2):
        a = oracle.roi_align_forward(x, rois, (14, 14), 1 / 16, sr, True)
        r = oracle.roi_align_rotated_forward(x, rrois, (14, 14), 1 / 16, sr)
        assert np.allclose(a, r, atol=1e-4)


def test_roi_align_grid_sample_equivalence_and_autograd():
    # test_roi_align.py:64-77 (grid_sample + avg_pool formulation) and backward vs autograd of it
    rng = np.random.default_rng(2)
    H = W = 30
    x = (rng.random((1, 2, H, W)) * 100).astype(np.float32)
    box = np.array([[0, 10, 10, 20, 20], [0, 3.3, 5.1, 17.2, 26.9]], np.float32)
    for ratio in (1, 2, 3):
        out = oracle.roi_align_forward(x, box, (5, 5), 1.0, ratio, True)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        outs = []
        for b in box:
            n = 5 * ratio
            t = (torch.arange(n, dtype=torch.float64) + 0.5) / n
            px = b[1] + t * (b[3] - b[1])
            py = b[2] + t * (b[4] - b[2])
            gx = px / W * 2 - 1
            gy = py / H * 2 - 1
            grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1)[None]
            s = torch.nn.functional.grid_sample(xt, grid, align_corners=False, padding_mode="border")
            outs.append(torch.nn.functional.avg_pool2d(s, ratio))
        ref_out = torch.cat(outs)
        assert np.allclose(out, ref_out.detach().numpy(), rtol=1e-5, atol=1e-4)
        g = rng.standard_normal(out.shape).astype(np.float32)
        ref_out.backward(torch.from_numpy(g).double())
        gin = oracle.roi_align_backward(g, box, x.shape, 1.0, ratio, True)
        assert np.allclose(gin, xt.grad.numpy(), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- ROIAlignRotated
def test_roi_align_rotated_known_answers():
    # /root/reference/tests/layers/test_roi_align_rotated.py:30-71
    x = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    exp = np.array([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0],
                    [12.0, 12.5, 13.0, 13.5]], np.float32)
    for k, ang in enumerate((0, 90, 180, 270)):
        rois = np.array([[0, 2, 2, 2, 2, ang]], np.float32)
        o = oracle.roi_align_rotated_forward(x, rois, (4, 4), 1.0, 0)[0, 0]
        assert np.allclose(o, np.rot90(exp, -k), atol=1e-4), ang  # :67-69 rotated CW


def test_roi_align_rotated_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "roi_align_rotated.npz"))
    N, C, H, W = d["x"].shape
    for sr in (0, 2):
        o = oracle.roi_align_rotated_forward(d["x"], d["rois"], (7, 7), 0.5, sr)
        assert np.array_equal(o, d[f"out_sr{sr}"])  # bit-exact vs compiled reference
        g = oracle.roi_align_rotated_backward(d["grad"], d["rois"], (N, C, H, W), 0.5, sr)
        assert np.array_equal(g, d[f"gin_sr{sr}"])


def test_roi_align_rotated_negative_size_raises():
    x = np.zeros((1, 1, 5, 5), np.float32)
    with pytest.raises(RuntimeError):
        oracle.roi_align_rotated_forward(x, np.array([[0, 2, 2, -1, 2, 0]], np.float32), (2, 2), 1.0, 0)


# ---------------------------------------------------------------- IoU
def test_pairwise_iou_known_answers():
    # /root/reference/tests/structures/test_boxes.py:152-186
    b1 = np.array([[0, 0, 1, 1], [0, 0, 1, 1]], np.float32)
    b2 = np.array([[0, 0, 1, 1], [0, 0.5, 1, 1], [0, 0, 0.5, 1], [0, 0, 0.5, 0.5],
                   [0.5, 0.5, 1, 1], [0.5, 0.5, 1.5, 1.5]], np.float32)
    iou = oracle.pairwise_iou(b1, b2)
    exp = np.array([[1, .5, .5, .25, .25, .25 / 1.75]] * 2, np.float32)
    assert np.allclose(iou, exp)
    ioa = oracle.pairwise_iou(b1, b2, "ioa")
    assert np.allclose(ioa, np.array([[1, 1, 1, 1, 1, .25]] * 2, np.float32))


def test_pairwise_iou_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "pairwise_iou.npz"))
    for mode in ("iou", "ioa", "intersection"):
        assert np.array_equal(oracle.pairwise_iou(d["b1"], d["b2"], mode), d[mode])


def test_box_iou_rotated_known_answers():
    # /root/reference/tests/structures/test_rotated_boxes.py
    f = lambda a, b: oracle.box_iou_rotated(np.array(a, np.float32), np.array(b, np.float32))
    assert np.allclose(f([[0.5, 0.5, 1, 1, 0]], [[0.25, 0.5, 0.5, 1, 0]]), 0.5)  # :46-51
    assert np.allclose(f([[565, 565, 10, 10.0, 0]], [[565, 565, 10, 8.3, 0]]), 0.83, atol=1e-5)  # :61-68
    # :277-290 45 degrees
    e = f([[1, 1, np.sqrt(2), np.sqrt(2), 0], [1, 1, 2, 2, 0]] * 1, [[1, 1, 2, 2, 45]])
    assert np.allclose(e[0, 0], 0.5, atol=1e-5)
    # :292-299 orthogonal
    assert np.allclose(f([[5, 5, 10, 6, 55]], [[5, 5, 10, 6, -35]]), 6.0 * 6 / (4 * 6 + 10 * 6), atol=1e-5)
    # :348-369 issue 1207 -> 0
    assert f([[160.0, 153.0, 230.0, 23.0, -37.0]], [[-0.122, 197.5, 0.122, 155.5, 90.0]])[0, 0] < 1e-4
    # :97-147 issues 2154 / 2167 -> 1
    a = [[296.6620178222656, 458.73883056640625, 23.515729904174805, 47.677001953125, 0.08795166015625]]
    b = [[296.66201781, 458.73882916, 23.51573, 47.67702, 0.087951]]
    assert np.allclose(f(a, b), 1.0, atol=1e-3)
    a = [[2563.74462890625000000000, 1436.79016113281250000000, 2174.70336914062500000000,
          214.09500122070312500000, 115.11834716796875000000]]
    assert np.allclose(f(a, a), 1.0, atol=1e-3)
    # :78-95 extreme coords >= 0
    assert f([[1e4, 1e4, 1e-3, 1e-3, 0]], [[1e4, 1e4, 1e3, 1e3, 30]])[0, 0] >= 0


def test_rotated_iou_nms_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "rotated_iou_nms.npz"))
    assert np.array_equal(oracle.box_iou_rotated(d["b1"], d["b2"]), d["iou"])
    for thr in (0.2, 0.5, 0.7):
        k = oracle.nms_rotated(d["nms_boxes"], d["nms_scores"], thr)
        assert np.array_equal(k, d[f"keep_{int(thr * 10)}"])


# ---------------------------------------------------------------- NMS
def _greedy_nms_python(boxes, scores, thr):
    """Restates the reference's own test oracle tests/layers/test_nms_rotated.py:44-66."""
    order = np.argsort(-scores, kind="stable")
    keep = []
    x1, y1, x2, y2 = boxes.T
    areas = (x2 - x1) * (y2 - y1)
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[1:][ovr <= thr]
    return np.array(keep, np.int64)


def _random_boxes(rng, n, size=100.0):
    b = rng.random((n, 4), dtype=np.float32) * np.float32(size * 0.5)
    b = np.maximum(b, 1.0)
    b[:, 2:] += b[:, :2]
    return b.astype(np.float32)


def test_nms_matches_reference_greedy_oracle():
    rng = np.random.default_rng(3)
    for n in (1, 2, 63, 64, 65, 500, 2000):
        b = _random_boxes(rng, n)
        s = (rng.permutation(n) / max(n, 1)).astype(np.float32)
        for thr in (0.2, 0.5, 0.8):
            assert np.array_equal(oracle.nms(b, s, thr), _greedy_nms_python(b, s, np.float32(thr) if False else thr))


def test_nms_rotated_zero_angle_equals_nms():
    # test_nms_rotated.py:100-112 (rotated at 0 deg vs horizontal NMS; reference allows edit
    # distance <= 1 because of >= vs >; with generic boxes they agree exactly)
    rng = np.random.default_rng(4)
    b = _random_boxes(rng, 300)
    s = (rng.permutation(300) / 300).astype(np.float32)
    rb = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1],
                   np.zeros(300, np.float32)], 1)
    for thr in (0.2, 0.5, 0.8):
        k1, k2 = oracle.nms(b, s, thr), oracle.nms_rotated(rb, s, thr)
        assert abs(len(k1) - len(k2)) <= 1 and len(set(k1) ^ set(k2)) <= 2


def test_batched_nms_is_per_class_nms():
    rng = np.random.default_rng(5)
    n = 1500
    b, s = _random_boxes(rng, n), (rng.permutation(n) / n).astype(np.float32)
    idx = rng.integers(0, 7, n)
    k = oracle.batched_nms(b, s, idx, 0.5)
    exp = np.concatenate([np.nonzero(idx == c)[0][oracle.nms(b[idx == c], s[idx == c], 0.5)] for c in range(7)])
    exp = exp[np.argsort(-s[exp], kind="stable")]
    assert np.array_equal(k, exp)
    assert len(oracle.batched_nms(np.zeros((0, 4)), np.zeros(0), np.zeros(0), 0.5)) == 0


# ---------------------------------------------------------------- paste_masks
def test_paste_masks_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "paste_masks.npz"))
    h, w = d["shape"]
    n = d["masks"].shape[0]
    exp = np.unpackbits(d["out_bits"])[: n * h * w].reshape(n, h, w).astype(bool)
    out = oracle.paste_masks_in_image(d["masks"], d["boxes"], (int(h), int(w)), 0.5)
    assert np.array_equal(out, exp)
    out8 = oracle.paste_masks_in_image(d["masks"], d["boxes"], (int(h), int(w)), -1)
    assert np.array_equal(out8, d["out_u8"])


def test_paste_masks_device_path_golden(golden_dir):
    """The oracle's skip_empty=False mode == the reference's `_do_paste_mask(..., skip_empty=False)` (what it runs
    for device tensors): the whole image is sampled, pixels up to half a mask pixel outside the box are non-zero."""
    d = np.load(os.path.join(golden_dir, "paste_masks_full.npz"))
    h, w = int(d["shape"][0]), int(d["shape"][1])
    n = d["masks"].shape[0]
    out8 = oracle.paste_masks_in_image(d["masks"], d["boxes"], (h, w), -1, skip_empty=False)
    assert np.array_equal(out8, d["out_u8"])
    for thr, key in ((0.1, "out_thr01"), (0.5, "out_thr05")):
        exp = np.unpackbits(d[key])[: n * h * w].reshape(n, h, w).astype(bool)
        assert np.array_equal(oracle.paste_masks_in_image(d["masks"], d["boxes"], (h, w), thr, skip_empty=False), exp)
    # ... and the CPU path (bbox region only) really differs from it below threshold 0.5
    cpu8 = oracle.paste_masks_in_image(d["masks"], d["boxes"], (h, w), -1, skip_empty=True)
    assert (cpu8 != out8).sum() > 50


# ---------------------------------------------------------------- deformable conv
DCN_GOLDEN = np.array([[30, 41.25, 48.75, 45, 28.75], [62.25, 81, 90, 80.25, 50.25],
                       [99.75, 126, 135, 117.75, 72.75], [105, 131.25, 138.75, 120, 73.75],
                       [71.75, 89.25, 93.75, 80.75, 49.5]], np.float32)


def test_deform_conv_golden():
    # /root/reference/tests/layers/test_deformable.py:16-58
    x = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    off = np.full((1, 18, 5, 5), 0.5, np.float32)
    w = np.ones((1, 1, 3, 3), np.float32)
    o = oracle.deform_conv_forward(x, off, w, padding=1)
    assert np.allclose(o[0, 0], DCN_GOLDEN)
    m = np.full((1, 9, 5, 5), 0.5, np.float32)
    o2 = oracle.deform_conv_forward(x, off, w, mask=m, padding=1)
    assert np.allclose(o2[0, 0], DCN_GOLDEN * 0.5)


@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("groups,dg", [(1, 1), (2, 2), (2, 1)])
def test_deform_conv_fwd_bwd_vs_torch_autograd(modulated, groups, dg):
    torch.manual_seed(7)
    B, C, H, W, Co = 2, 4, 7, 9, 6
    stride, pad, dil = (1, 1), (1, 1), (1, 1)
    x = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    off = (torch.randn(B, dg * 18, H, W, dtype=torch.float64) * 1.5).requires_grad_(True)
    msk = torch.sigmoid(torch.randn(B, dg * 9, H, W, dtype=torch.float64)).requires_grad_(True) if modulated else None
    wt = (torch.randn(Co, C // groups, 3, 3, dtype=torch.float64) * 0.2).requires_grad_(True)
    bias = torch.randn(Co, dtype=torch.float64, requires_grad=True) if modulated else None
    out = dcn_torch(x, off, wt, msk, bias, stride, pad, dil, groups, dg)
    go = torch.randn_like(out)
    out.backward(go)
    f = lambda t: t.detach().float().numpy() if t is not None else None
    o = oracle.deform_conv_forward(f(x), f(off), f(wt), mask=f(msk), bias=f(bias), stride=stride,
                                   padding=pad, dilation=dil, groups=groups, deformable_groups=dg)
    assert np.allclose(o, f(out), rtol=1e-4, atol=1e-4)
    g = oracle.deform_conv_backward(f(x), f(off), f(wt), f(go), mask=f(msk), with_bias=modulated,
                                    stride=stride, padding=pad, dilation=dil, groups=groups,
                                    deformable_groups=dg)
    assert np.allclose(g["grad_input"], f(x.grad), rtol=1e-4, atol=1e-4)
    assert np.allclose(g["grad_weight"], f(wt.grad), rtol=1e-4, atol=1e-4)
    # d/d(offset) of a bilinear sample is discontinuous on integer coordinates; random offsets
    # are generic so the analytic formulas must agree
    assert np.allclose(g["grad_offset"], f(off.grad), rtol=1e-3, atol=1e-3)
    if modulated:
        assert np.allclose(g["grad_mask"], f(msk.grad), rtol=1e-4, atol=1e-4)
        assert np.allclose(g["grad_bias"], f(bias.grad), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- live compiled reference
def test_live_against_compiled_reference():
    from conftest import need_reference

    need_reference(ref.have_compiled() or ref.have_tree(), "oracle/_ref/libd2ref.so (compiled reference)")
    ops = ref.compiled()
    rng = np.random.default_rng(11)
    n = 200
    b = np.stack([rng.uniform(0, 80, n), rng.uniform(0, 80, n), rng.uniform(1, 50, n),
                  rng.uniform(1, 50, n), rng.uniform(-180, 180, n)], 1).astype(np.float32)
    iou = ops.box_iou_rotated(torch.from_numpy(b), torch.from_numpy(b[:50])).numpy()
    assert np.array_equal(oracle.box_iou_rotated(b, b[:50]), iou)
    s = (rng.permutation(n) / n).astype(np.float32)
    k = ops.nms_rotated(torch.from_numpy(b), torch.from_numpy(s), 0.3).numpy()
    assert np.array_equal(oracle.nms_rotated(b, s, 0.3), k)


# ---------------------------------------------------------------------------------------- Matcher
def _matcher_cases(g):
    for name in ("rpn", "roi", "retina", "three"):
        cfg = g[f"{name}_cfg"]
        t = int(cfg[0])
        thr, lab, low = list(cfg[1:1 + t]), [int(v) for v in cfg[1 + t:2 + 2 * t]], bool(cfg[-1])
        yield name, thr, lab, low


def test_matcher_restatement_matches_reference_class(golden_dir):
    """oracle.matcher == the reference's Matcher (detectron2/modeling/matcher.py) on the reference's own
    pairwise_iou matrix: first-maximum ties, threshold intervals, low-quality matches incl. a ground
    truth whose row maximum is 0, and the empty-ground-truth path."""
    g = np.load(os.path.join(golden_dir, "matcher.npz"))
    for name, thr, lab, low in _matcher_cases(g):
        m, l = oracle.matcher(g["quality"], thr, lab, low)
        assert np.array_equal(m, g[f"{name}_matches"]), name
        assert np.array_equal(l, g[f"{name}_labels"]), name
        assert l.dtype == np.int8 and m.dtype == np.int64
    m, l = oracle.matcher(np.zeros((0, 7), np.float32), [0.3, 0.7], [0, -1, 1], True)
    assert np.array_equal(m, g["empty_matches"]) and np.array_equal(l, g["empty_labels"])
    # and the restated IoU feeds it bit-identically
    assert np.array_equal(oracle.pairwise_iou(g["gt"], g["boxes"]), g["quality"])


# ---------------------------------------------------------------------------------------- RPN proposals
def _rpn_case(g):
    L = 3
    anchors = [g[f"anchors{l}"] for l in range(L)]
    logits = [g[f"logits{l}"] for l in range(L)]
    deltas = [g[f"deltas{l}"] for l in range(L)]
    n = logits[0].shape[0]
    hw = [tuple(int(v) for v in g["image_hw"])] * n
    return anchors, logits, deltas, hw


def test_rpn_restatement_matches_reference_functions(golden_dir):
    """oracle/rpn.py vs the reference's Box2BoxTransform.apply_deltas and find_top_rpn_proposals
    (tests/golden/rpn_proposals.npz): decode to ~1 ulp of exp(), selection / NMS result identical."""
    from oracle import rpn
    g = np.load(os.path.join(golden_dir, "rpn_proposals.npz"))
    anchors, logits, deltas, hw = _rpn_case(g)
    for l in range(3):
        for i in range(logits[0].shape[0]):
            got = rpn.apply_deltas(deltas[l][i], anchors[l])
            assert np.allclose(got, g[f"decoded{l}"][i], rtol=2e-6, atol=2e-4)
    res = rpn.find_top_rpn_proposals(anchors, logits, deltas, hw, float(g["nms_thresh"]), int(g["pre_nms_topk"]),
                                     int(g["post_nms_topk"]), float(g["min_box_size"]))
    for i, (b, s) in enumerate(res):
        assert np.array_equal(s, g[f"scores_img{i}"])          # same proposals, same order
        assert np.allclose(b, g[f"boxes_img{i}"], rtol=2e-6, atol=2e-4)


def test_mask_head_restatement_matches_reference_functions(golden_dir):
    """oracle/mask_head.py vs the reference's own mask_rcnn_loss / mask_rcnn_inference run on CPU
    (tests/golden/mask_head.npz, generated through oracle/ref.py::py_mask_head): loss, the logged accuracy /
    false positive / false negative, autograd gradient (x 1.75 upstream) and inference probabilities."""
    from oracle import mask_head as omh

    g = np.load(os.path.join(golden_dir, "mask_head.npz"))
    for name in ("a", "b", "agn"):
        x, cls, gt = g[f"{name}_logits"], g[f"{name}_classes"], g[f"{name}_gt"]
        loss, st = omh.mask_rcnn_loss(x, cls, gt)
        assert abs(loss - float(g[f"{name}_loss"])) <= 1e-5 * abs(float(g[f"{name}_loss"]))
        for k in ("accuracy", "false_positive", "false_negative"):
            assert abs(st[k] - float(g[f"{name}_{k}"])) < 1e-12, k
        grad = omh.mask_rcnn_loss_grad(x, cls, gt, 1.75)
        assert np.abs(grad - g[f"{name}_grad_x1p75"]).max() <= 1e-6 * np.abs(grad).max() + 1e-12
        assert np.abs(grad[g[f"{name}_grad_x1p75"] == 0]).max(initial=0.0) < 1e-30  # other planes: zero (fp32 saturates at -90)
        probs = omh.mask_rcnn_inference(x, cls)
        assert probs.shape == g[f"{name}_probs"].shape
        assert np.abs(probs - g[f"{name}_probs"]).max() <= 1e-6


def test_dense_detector_restatement_matches_reference_functions(golden_dir):
    """oracle/dense_detector.py vs the reference's own DenseDetector._decode_multi_level_predictions run on CPU
    (tests/golden/dense_detector.npz, oracle/ref.py::py_dense_detector).  Case 1 (no equal scores among the
    candidates): same (anchor, class) selection in the same order, scores and decoded boxes to the rounding of exp()."""
    from oracle import dense_detector as odd

    g = np.load(os.path.join(golden_dir, "dense_detector.npz"))
    anchors = [g[f"anchors{l}"] for l in range(3)]
    for i in range(2):
        b, s, c = odd.decode_multi_level(anchors, [g[f"logits{l}"][i] for l in range(3)],
                                         [g[f"deltas{l}"][i] for l in range(3)], float(g["score_thresh"]),
                                         int(g["topk"]), tuple(g["weights"]))
        assert np.array_equal(c, g[f"classes_img{i}"])
        np.testing.assert_allclose(s, g[f"scores_img{i}"], rtol=2e-6, atol=0)
        np.testing.assert_allclose(b, g[f"boxes_img{i}"], rtol=2e-6, atol=1e-4)


def test_dense_detector_logit_ranking_refines_reference_topk(golden_dir):
    """Case 2 ("t_" keys): heavy ties -- quantised logits, saturated sigmoids (different logits, one fp32 score),
    +0 / -0, tied k-th scores.  The logit ranking must select what the reference's `topk` selected and order it like
    the reference, except INSIDE groups of equal fp32 score (unspecified for torch.topk)."""
    from _dense_compare import assert_same_selection_up_to_ties
    from oracle import dense_detector as odd

    g = np.load(os.path.join(golden_dir, "dense_detector.npz"))
    anchors = [g[f"t_anchors{l}"] for l in range(3)]
    thr, topk = float(g["t_score_thresh"]), int(g["t_topk"])
    tot_id = tot_tail = tot_moved = 0
    for i in range(2):
        got = odd.decode_multi_level(anchors, [g[f"t_logits{l}"][i] for l in range(3)],
                                     [g[f"t_deltas{l}"][i] for l in range(3)], thr, topk, tuple(g["weights"]))
        ref = (g[f"t_boxes_img{i}"], g[f"t_scores_img{i}"], g[f"t_classes_img{i}"])
        n_id, n_tail, n_moved = assert_same_selection_up_to_ties(got, ref, g[f"t_counts_img{i}"], topk)
        tot_id, tot_tail, tot_moved = tot_id + n_id, tot_tail + n_tail, tot_moved + n_moved
    # the fixture really exercises ties (rows sit at other positions than in the reference) and most rows are pinned
    assert tot_moved > 20 and tot_id > 4 * tot_tail, (tot_id, tot_tail, tot_moved)


def test_logit_lower_bound_is_the_exact_threshold():
    """`sigmoid(x) > t` in exact arithmetic <=> x > log(t / (1 - t)) <=> x >= logit_lower_bound(t): the fp32 bound and
    its predecessor straddle the logit of t evaluated in extended precision; plus the special thresholds."""
    from oracle import dense_detector as odd

    for t in [0.05, 0.3, 0.5, 0.2, 1e-6, 0.999999, 0.9]:
        t32 = np.float32(t)
        b = odd.logit_lower_bound(t)
        below = np.nextafter(b, np.float32(-np.inf), dtype=np.float32)
        t_ld = np.longdouble(t32)
        logit_t = np.log(t_ld / (np.longdouble(1) - t_ld))
        assert np.longdouble(b) > logit_t and not np.longdouble(below) > logit_t, t
    assert np.isnan(odd.logit_lower_bound(1.0)) and np.isnan(odd.logit_lower_bound(1.5))
    assert odd.logit_lower_bound(-0.1) == -np.inf
    assert odd.sigmoid32(odd.logit_lower_bound(0.0)) > 0


def test_polygon_rasteriser_restatement_known_answers():
    """oracle.polygons_to_bitmask (cocoapi rleFrPoly restated; pycocotools itself is not available: parity unpinned)
    against the reference's own known answer -- tests/structures/test_masks.py:31-38: the polygon of an integer box,
    rasterised on a 4 x 4 grid, has exactly that box as its bounding box (here: fills [x0, x1) x [y0, y1)) -- and
    against structural properties of the algorithm (union of polygons, translation by whole pixels)."""
    for box in ([1, 0, 4, 4], [1, 1, 3, 4], [0, 0, 0, 0]):
        b = np.array(box, np.float64)
        m = oracle.polygons_to_bitmask([b[[0, 1, 2, 1, 2, 3, 0, 3]]], 4, 4)
        exp = np.zeros((4, 4), bool)
        exp[box[1]:box[3], box[0]:box[2]] = True
        assert np.array_equal(m, exp)
    rng = np.random.default_rng(0)
    a = rng.uniform(2, 26, 14)
    b2 = rng.uniform(2, 26, 10)
    ma, mb = oracle.polygons_to_bitmask([a], 32, 32), oracle.polygons_to_bitmask([b2], 32, 32)
    assert np.array_equal(oracle.polygons_to_bitmask([a, b2], 32, 32), ma | mb)        # rleMerge = union
    shifted = a.copy()
    shifted[0::2] += 3
    shifted[1::2] += 2
    ms = oracle.polygons_to_bitmask([shifted], 32, 32)
    assert np.array_equal(ms[2:, 3:], ma[:-2, :-3]) and ma.sum() > 20                   # whole-pixel translation
    # rasterize_polygons_within_box: a box equal to the frame is the identity transform
    assert np.array_equal(oracle.rasterize_polygons_within_box([a], [0, 0, 32, 32], 32), ma)


# ---------------------------------------------------------------- DCN: the restatement pinned to the reference's kernels
def test_deform_conv_forward_and_backward_vs_reference_gpu_goldens(golden_dir):
    """SURVEY 8(c) called DCN backward "parity-unpinned" (no CPU implementation, no gradient test upstream).  The golden
    file holds what the reference's OWN kernels (csrc/deformable/*.cu compiled as HIP by oracle/build_ref.py:build_dcn,
    driven by the reference's own layers/deform_conv.py) produced on an MI355X for the cases of tests/_dcn_cases.py
    (tests/golden/make_dcn_reference_gpu.py).  The C restatement -- oracle.deform_conv_forward / _backward, what every
    other DCN test compares the HIP kernels with -- must reproduce every element of every tensor: out, grad_input,
    grad_offset, grad_mask, grad_weight, grad_bias; v1 and v2; conv groups, deformable groups, stride, dilation,
    padding, 5x5 taps, far offsets.  Bound per element: 1e-4 |ref| + 1e-6 max|ref|."""
    import _dcn_cases as dc
    from conftest import assert_close_fp32

    g = np.load(os.path.join(golden_dir, "dcn_reference_gpu.npz"))
    for name in dc.SMALL:
        case = dc.make_small(name)
        cs = dc.input_checksum(case)
        assert abs(float(g[f"small/{name}/checksum"]) - cs) < 1e-6 * cs, "seeded inputs differ from the golden file's"
        f = lambda t: None if t is None else t.numpy()
        kw = case["kw"]
        out = oracle.deform_conv_forward(f(case["x"]), f(case["offset"]), f(case["weight"]), mask=f(case["mask"]),
                                         bias=f(case["bias"]), **kw)
        assert_close_fp32(out, g[f"small/{name}/out"], f"oracle_dcn/{name}/out")
        grads = oracle.deform_conv_backward(f(case["x"]), f(case["offset"]), f(case["weight"]), f(case["grad_out"]),
                                            mask=f(case["mask"]), with_bias=case["mask"] is not None, **kw)
        for k, v in grads.items():
            if v is not None:
                assert_close_fp32(v, g[f"small/{name}/{k}"], f"oracle_dcn/{name}/{k}")


def test_fp32_roi_align_distance_from_fp64():
    """Evidence for tests/conftest.py: ROI_FLOOR.  The reference's ROIAlign arithmetic in fp32 (the oracle, operation
    for operation as ROIAlignRotated_cpu.cpp:27-129 at angle 0) against an fp64 evaluation of the same fp32 ROIs: the
    sample coordinates are fp32 products / sums of ~100-px numbers, so the fp32 result sits ~1e-5 max|y| from the
    exact one.  No fp32 implementation with a different (equally valid) rounding order can be held closer to the oracle
    than the oracle is to the truth: the per-element floor for ROIAlign is 1e-5 max|ref|, not 1e-6."""
    rng = np.random.default_rng(436)
    H, W, R, g, scale = 128, 160, 14, 2, 0.25
    f = rng.standard_normal((1, 4, H, W)).astype(np.float32)
    s = np.exp(rng.uniform(np.log(8), np.log(400), 24))
    cx, cy = rng.uniform(0, 640, 24), rng.uniform(0, 512, 24)
    rois = np.stack([np.zeros(24), cx - s / 2, cy - s / 2, cx + s / 2, cy + s / 2], 1).astype(np.float32)
    got = oracle.roi_align_forward(f, rois, (R, R), scale, g, True)
    f64 = f.astype(np.float64)
    exp = np.zeros((24, 4, R, R))
    for k, (b, x1, y1, x2, y2) in enumerate(rois.astype(np.float64)):
        sx, sy = x1 * scale - 0.5, y1 * scale - 0.5
        bw, bh = (x2 * scale - x1 * scale) / R, (y2 * scale - y1 * scale) / R
        for ph in range(R):
            for pw in range(R):
                acc = np.zeros(4)
                for iy in range(g):
                    y = sy + ph * bh + (iy + 0.5) * bh / g
                    for ix in range(g):
                        x = sx + pw * bw + (ix + 0.5) * bw / g
                        if y < -1 or y > H or x < -1 or x > W:
                            continue
                        yy, xx = max(y, 0.0), max(x, 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        acc += ((1 - ly) * (1 - lx) * f64[0, :, yl, xl] + (1 - ly) * lx * f64[0, :, yl, xh] +
                                ly * (1 - lx) * f64[0, :, yh, xl] + ly * lx * f64[0, :, yh, xh])
                exp[k, :, ph, pw] = acc / (g * g)
    d = np.abs(got - exp).max() / np.abs(exp).max()
    assert 1e-6 < d < 1e-4, d  # ~1e-5: far above the 1e-6 a pure summation-order difference would give


@pytest.mark.parametrize("case", ["maskrcnn", "agnostic", "ragged", "nonfinite", "all_pass", "none_pass", "ties"])
def test_fast_rcnn_inference_vs_reference_source(case):
    """oracle/fast_rcnn.py against the reference's OWN roi_heads/fast_rcnn.py (fast_rcnn_inference_single_image, loaded
    unmodified from the staged bytecode) on the CPU; `detectron2.layers.batched_nms` is torchvision's upstream -- not
    installed here -- so both sides use the oracle's port of it (pinned above to the reference's known answers): what
    this pins is everything around the NMS (finite-row mask, clip, threshold, nonzero order, the three gathers)."""
    import torch
    from _fast_rcnn_cases import make
    from conftest import need_reference
    from oracle import fast_rcnn as ofr
    from oracle import ref

    need_reference(ref.have_py(), "oracle/_ref/py (the reference's fast_rcnn.py)")

    def bnms(b, s, i, t):
        return torch.from_numpy(oracle.batched_nms(b.numpy(), s.numpy(), i.numpy().astype(np.int64), t))

    m = ref.py_fast_rcnn(bnms)
    boxes, scores, shapes, thr, nms, topk = make(case)
    for i in range(len(boxes)):
        inst, kept = m.fast_rcnn_inference_single_image(torch.from_numpy(boxes[i]), torch.from_numpy(scores[i]), shapes[i],
                                                        thr, nms, topk)
        wb, ws, wc, wr = ofr.fast_rcnn_inference_single_image(boxes[i], scores[i], shapes[i], thr, nms, topk)
        assert np.array_equal(inst.pred_boxes.tensor.numpy(), wb)
        assert np.array_equal(inst.scores.numpy(), ws)
        assert np.array_equal(inst.pred_classes.numpy(), wc)
        assert np.array_equal(kept.numpy(), wr)
