"""GPU parity of the fused anchor / proposal matcher (detectron2_amd/csrc/matcher.hip):
bit-exact against the reference's Matcher on the reference's pairwise_iou (tests/golden/matcher.npz) and
against the oracle at the RPN's full size (16 x 268,569 anchors)."""
import os

import numpy as np
import pytest
import torch

import oracle
from detectron2_amd.modeling import Matcher
from detectron2_amd.structures import Boxes, pairwise_iou

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cases(g):
    for name in ("rpn", "roi", "retina", "three"):
        cfg = g[f"{name}_cfg"]
        t = int(cfg[0])
        yield name, list(cfg[1:1 + t]), [int(v) for v in cfg[1 + t:2 + 2 * t]], bool(cfg[-1])


def test_matcher_golden_fused_and_matrix(golden_dir):
    g = np.load(os.path.join(golden_dir, "matcher.npz"))
    gt, boxes = torch.from_numpy(g["gt"]).to(DEV), torch.from_numpy(g["boxes"]).to(DEV)
    q = torch.from_numpy(g["quality"]).to(DEV)
    for name, thr, lab, low in _cases(g):
        mt = Matcher(thr, lab, allow_low_quality_matches=low)
        for m, l in (mt.match_boxes(Boxes(gt), Boxes(boxes)), mt(q), mt(pairwise_iou(Boxes(gt), Boxes(boxes)))):
            assert m.dtype == torch.int64 and l.dtype == torch.int8
            assert np.array_equal(m.cpu().numpy(), g[f"{name}_matches"]), name
            assert np.array_equal(l.cpu().numpy(), g[f"{name}_labels"]), name


def test_matcher_empty_inputs():
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    m, l = mt.match_boxes(torch.zeros(0, 4, device=DEV), torch.rand(7, 4, device=DEV))
    assert m.tolist() == [0] * 7 and l.tolist() == [0] * 7 and l.dtype == torch.int8
    m, l = mt(torch.zeros(0, 7, device=DEV))
    assert m.tolist() == [0] * 7 and l.tolist() == [0] * 7
    m, l = mt.match_boxes(torch.rand(3, 4, device=DEV), torch.zeros(0, 4, device=DEV))
    assert m.numel() == 0 and l.numel() == 0


def test_matcher_many_ground_truth_chunks():
    """M = 1,300 ground-truth boxes: three LDS passes over the ground truth per block."""
    rng = np.random.default_rng(5)
    g = rng.uniform(0, 500, (1300, 4)).astype(np.float32)
    g[:, 2:] = g[:, :2] + rng.uniform(5, 100, (1300, 2)).astype(np.float32)
    a = rng.uniform(0, 500, (5000, 4)).astype(np.float32)
    a[:, 2:] = a[:, :2] + rng.uniform(5, 100, (5000, 2)).astype(np.float32)
    a[:50] = g[600:650]
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    m, l = mt.match_boxes(torch.from_numpy(g).to(DEV), torch.from_numpy(a).to(DEV))
    em, el = oracle.matcher(oracle.pairwise_iou(g, a), [0.3, 0.7], [0, -1, 1], True)
    assert np.array_equal(m.cpu().numpy(), em) and np.array_equal(l.cpu().numpy(), el)


def test_matcher_rpn_full_size():
    """BASELINE configs[1]: 16 ground-truth boxes x 268,569 anchors, RPN thresholds (0.3, 0.7) with
    low-quality matches; also the ROI-head case 16 x 1,016 at 0.5."""
    import bench
    w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
    gt, an = w.gt[0], w.anchors
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    m, l = mt.match_boxes(gt, an)
    em, el = oracle.matcher(oracle.pairwise_iou(gt.cpu().numpy(), an.cpu().numpy()), [0.3, 0.7], [0, -1, 1], True)
    assert np.array_equal(m.cpu().numpy(), em) and np.array_equal(l.cpu().numpy(), el)
    assert (el == 1).sum() > 0 and (el == -1).sum() > 0
    mr = Matcher([0.5], [0, 1], allow_low_quality_matches=False)
    m, l = mr.match_boxes(gt, w.props[0])
    em, el = oracle.matcher(oracle.pairwise_iou(gt.cpu().numpy(), w.props[0].cpu().numpy()), [0.5], [0, 1], False)
    assert np.array_equal(m.cpu().numpy(), em) and np.array_equal(l.cpu().numpy(), el)


@pytest.mark.parametrize("low", [True, False])
def test_match_boxes_batch_equals_the_per_image_loop(low):
    """d2amd_match_boxes_batch: the ground truth of several images (different counts, one image without any, more images
    than one launch holds) against the SAME boxes == the reference's per-image loop (rpn.py:331-353), bit for bit,
    and == the oracle."""
    rng = np.random.default_rng(17 + low)
    n = 20000
    a = rng.uniform(0, 600, (n, 4)).astype(np.float32)
    a[:, 2:] = a[:, :2] + rng.uniform(4, 200, (n, 2)).astype(np.float32)
    counts = [5, 0, 40, 1, 17] + [3] * 14  # 19 images: two launches
    gts = []
    for i, m in enumerate(counts):
        g = rng.uniform(0, 600, (m, 4)).astype(np.float32)
        g[:, 2:] = g[:, :2] + rng.uniform(10, 250, (m, 2)).astype(np.float32)
        if m:
            g[0] = a[100 + i]  # an exact match, shared by a row maximum
        gts.append(g)
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=low)
    at = torch.from_numpy(a).to(DEV)
    m, l = mt.match_boxes_batch([torch.from_numpy(g).to(DEV) for g in gts], at)
    assert tuple(m.shape) == tuple(l.shape) == (len(counts), n) and m.dtype == torch.int64 and l.dtype == torch.int8
    for i, g in enumerate(gts):
        em, el = mt.match_boxes(torch.from_numpy(g).to(DEV).reshape(-1, 4), at)
        assert torch.equal(m[i], em) and torch.equal(l[i], el), i
        if i < 5:
            om, ol = oracle.matcher(oracle.pairwise_iou(g, a), [0.3, 0.7], [0, -1, 1], low)
            assert np.array_equal(m[i].cpu().numpy(), om) and np.array_equal(l[i].cpu().numpy(), ol), i
    e = mt.match_boxes_batch([], at)
    assert tuple(e[0].shape) == (0, n)


def test_reference_known_answer():
    """/root/reference/tests/modeling/test_matcher.py:11-24, eager half (its TorchScript half is export: out of scope):
    the RPN matcher of config/defaults.py (IOU_THRESHOLDS [0.3, 0.7], IOU_LABELS [0, -1, 1], low-quality matches) on the
    test's 3 x 4 quality matrix."""
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    q = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]], device=DEV)
    matches, labels = mt(q)
    assert matches.tolist() == [1, 1, 2, 0] and matches.dtype == torch.int64
    assert labels.tolist() == [-1, 1, 0, 1] and labels.dtype == torch.int8
