"""GPU parity of the dense-detector (RetinaNet) selection path (csrc/topk.hip radix select + dense_decode_kernel,
SURVEY 8(f) row 2) through the C ABI: against the reference's own DenseDetector methods (tests/golden/
dense_detector.npz) and against the numpy restatement oracle/dense_detector.py at RetinaNet scale (N x 16 M scores).
Bars: the selected (anchor, class) pairs and their order exact (ties: lower flattened index first); scores / boxes
to the rounding of exp() (rtol 2e-6); final NMS result exact."""
import os

import numpy as np
import pytest
import torch

from detectron2_amd.modeling import dense_detector_inference_fused, dense_select_predictions, rpn_select_proposals
from oracle import dense_detector as odd

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _rows(boxes, scores, classes, valid, counts, i, sizes_k):
    """valid rows of image i, level by level (what the reference concatenates)."""
    b, s, c = [], [], []
    o = 0
    for l, kl in enumerate(sizes_k):
        n = int(counts[i, l])
        assert bool(valid[i, o:o + n].all()) and not bool(valid[i, o + n:o + kl].any())
        assert bool(torch.isinf(scores[i, o + n:o + kl]).all())
        b.append(boxes[i, o:o + n]); s.append(scores[i, o:o + n]); c.append(classes[i, o:o + n])
        o += kl
    return torch.cat(b).cpu().numpy(), torch.cat(s).cpu().numpy(), torch.cat(c).cpu().numpy()


def test_dense_select_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "dense_detector.npz"))
    anchors = [g[f"anchors{l}"] for l in range(3)]
    logits = [g[f"logits{l}"] for l in range(3)]
    deltas = [g[f"deltas{l}"] for l in range(3)]
    thr, topk, w = float(g["score_thresh"]), int(g["topk"]), tuple(g["weights"])
    out = dense_select_predictions([cu(a) for a in anchors], [cu(x) for x in logits], [cu(x) for x in deltas], thr, topk, w)
    boxes, scores, classes, valid, counts = out
    sizes_k = [min(a.shape[0] * 5, topk) for a in anchors]
    assert boxes.shape == (2, sum(sizes_k), 4) and classes.dtype == torch.int64 and counts.dtype == torch.int32
    for i in range(2):
        b, s, c = _rows(boxes, scores, classes, valid, counts.cpu(), i, sizes_k)
        assert np.array_equal(c, g[f"classes_img{i}"])
        np.testing.assert_allclose(s, g[f"scores_img{i}"], rtol=2e-6, atol=0)
        np.testing.assert_allclose(b, g[f"boxes_img{i}"], rtol=2e-6, atol=1e-4)


@pytest.mark.parametrize("case", ["retinanet_800x1344", "small_ragged", "all_pass", "none_pass", "big_k"])
def test_dense_select_vs_oracle(case):
    rng = np.random.default_rng(hash(case) % 1000)
    if case == "retinanet_800x1344":  # BASELINE configs[3] shapes: 9 anchors / location, 80 classes, 2 images
        N, K, sizes, thr, topk = 2, 80, [9 * 16800, 9 * 4200, 9 * 1050, 9 * 273, 9 * 77], 0.05, 1000
        mean = -4.0
    elif case == "small_ragged":
        N, K, sizes, thr, topk, mean = 3, 7, [1000, 1, 0, 37], 0.2, 64, -1.0
    elif case == "big_k":  # 10,000 selected per segment: 16,384-entry LDS sort (128 KB of dynamic LDS)
        N, K, sizes, thr, topk, mean = 2, 4, [3000, 2600], 0.0, 10000, 0.0
    elif case == "all_pass":
        N, K, sizes, thr, topk, mean = 1, 3, [5000, 10], 0.0, 100, 2.0
    else:
        N, K, sizes, thr, topk, mean = 2, 4, [300, 20], 0.999999, 50, -3.0
    anchors, logits, deltas = [], [], []
    for li, a_l in enumerate(sizes):
        c = rng.uniform(0, [1344, 800], (a_l, 2))
        wh = 32.0 * 2 ** li * np.exp(rng.uniform(-0.4, 0.4, (a_l, 2)))
        anchors.append(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32))
        logits.append((rng.standard_normal((N, a_l, K)) * 1.2 + mean).astype(np.float32))
        deltas.append((rng.standard_normal((N, a_l, 4)) * [0.2, 0.2, 0.3, 0.3]).astype(np.float32))
    boxes, scores, classes, valid, counts = dense_select_predictions(
        [cu(a) for a in anchors], [cu(x) for x in logits], [cu(x) for x in deltas], thr, topk)
    sizes_k = [min(a * K, topk) for a in sizes]
    counts = counts.cpu()
    for i in range(N):
        b, s, c = _rows(boxes, scores, classes, valid, counts, i, sizes_k)
        wb, ws, wc = odd.decode_multi_level(anchors, [x[i] for x in logits], [x[i] for x in deltas], thr, topk)
        assert len(s) == len(ws)
        # device exp() may differ from numpy's in the last bit: the ORDER can differ only between scores that close
        if not np.array_equal(c, wc):
            o1, o2 = np.lexsort((c, -s.astype(np.float64))), np.lexsort((wc, -ws.astype(np.float64)))
            np.testing.assert_allclose(s[o1], ws[o2], rtol=4e-6)
        else:
            np.testing.assert_allclose(s, ws, rtol=2e-6, atol=0)
            np.testing.assert_allclose(b, wb, rtol=2e-6, atol=1e-3)
    if case == "none_pass":
        assert int(counts.sum()) == 0


def test_topk_ties_resolve_to_lower_index_and_are_deterministic():
    """Heavy ties: quantised logits -> many equal scores at the k-th value; the selection must take the lowest
    flattened indices (ordered tie path of csrc/topk.hip) and be identical run to run."""
    rng = np.random.default_rng(5)
    N, K, sizes, thr, topk = 2, 4, [6000, 700], 0.1, 500
    anchors = [np.tile(np.array([[0, 0, 32, 32]], np.float32), (a, 1)) for a in sizes]
    logits = [np.round(rng.standard_normal((N, a, K)) * 2).astype(np.float32) for a in sizes]  # ~9 distinct values
    logits[1][1] = 3.0  # one segment where every element ties
    deltas = [np.zeros((N, a, 4), np.float32) for a in sizes]
    args = ([cu(a) for a in anchors], [cu(x) for x in logits], [cu(x) for x in deltas], thr, topk)
    r1 = dense_select_predictions(*args)
    r2 = dense_select_predictions(*args)
    for a, b in zip(r1, r2):
        assert torch.equal(a, b)
    boxes, scores, classes, valid, counts = r1
    sizes_k = [min(a * K, topk) for a in sizes]
    for i in range(N):
        _, s, c = _rows(boxes, scores, classes, valid, counts.cpu(), i, sizes_k)
        _, ws, wc = odd.decode_multi_level(anchors, [x[i] for x in logits], [x[i] for x in deltas], thr, topk)
        np.testing.assert_allclose(s, ws, rtol=2e-6)
        assert np.array_equal(c, wc)  # equal scores: lower flattened index first, exactly like the restatement
    # the same ordered-tie machinery behind the RPN selection: all logits equal -> the first k anchors of each level
    n_a = 5000
    an = cu(np.tile(np.array([[0, 0, 64, 64]], np.float32), (n_a, 1)))
    lg = torch.zeros(1, n_a, device=DEV)
    dl = torch.zeros(1, n_a, 4, device=DEV)
    dl[0, :, 0] = torch.arange(n_a, device=DEV) * 1e-3  # distinguishes the anchors in the decoded boxes
    b, s, v, _, _ = rpn_select_proposals([an], [lg], [dl], [(4000, 4000)], 300, 0.0)
    want = 0.0 + 64 * (torch.arange(300, device=DEV) * 1e-3)
    assert torch.allclose(b[0, :, 0], want, atol=1e-4) and bool(v.all())


def test_dense_detector_inference_fused_matches_oracle():
    rng = np.random.default_rng(11)
    N, K, sizes, thr, topk, nms_thr, max_det = 2, 6, [1200, 300, 75], 0.3, 200, 0.5, 50
    anchors, logits, deltas = [], [], []
    for li, a_l in enumerate(sizes):
        c = rng.uniform(0, [320, 256], (a_l, 2))
        wh = 24.0 * 2 ** li * np.exp(rng.uniform(-0.3, 0.3, (a_l, 2)))
        anchors.append(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32))
        logits.append((rng.standard_normal((N, a_l, K)) * 1.5 - 1.0).astype(np.float32))
        deltas.append((rng.standard_normal((N, a_l, 4)) * 0.1).astype(np.float32))
    res = dense_detector_inference_fused([cu(a) for a in anchors], [cu(x) for x in logits], [cu(x) for x in deltas],
                                         [(256, 320)] * N, thr, topk, nms_thr, max_det)
    for i, r in enumerate(res):
        wb, ws, wc = odd.inference_single_image(anchors, [x[i] for x in logits], [x[i] for x in deltas], thr, topk,
                                                nms_thr, max_det)
        assert len(r) == len(ws) <= max_det and r.image_size == (256, 320)
        assert np.array_equal(r.pred_classes.cpu().numpy(), wc)
        np.testing.assert_allclose(r.scores.cpu().numpy(), ws, rtol=2e-6)
        np.testing.assert_allclose(r.pred_boxes.tensor.cpu().numpy(), wb, rtol=2e-6, atol=1e-3)


def test_dense_select_errors():
    a = [torch.zeros(4, 4, device=DEV)]
    with pytest.raises(RuntimeError):  # topk beyond the LDS ordering limit
        dense_select_predictions(a, [torch.zeros(1, 4, 2, device=DEV)], [torch.zeros(1, 4, 4, device=DEV)], 0.05, 20000)
    with pytest.raises(NotImplementedError):  # CPU tensors: no fallback
        dense_select_predictions([torch.zeros(4, 4)], [torch.zeros(1, 4, 2)], [torch.zeros(1, 4, 4)], 0.05, 10)
    b, s, c, v, n = dense_select_predictions(a, [torch.zeros(0, 4, 2, device=DEV)], [torch.zeros(0, 4, 4, device=DEV)], 0.05, 10)
    assert b.shape == (0, 8, 4) and n.shape == (0, 1)
