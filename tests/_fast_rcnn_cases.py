"""Seeded inputs of the box-head inference tests (fast_rcnn.py:118-170): predict_boxes / predict_probs shaped arrays."""
import numpy as np


def make(case, seed=0):
    """-> (boxes list [R_i, Kb * 4], scores list [R_i, K + 1], image_shapes, score_thresh, nms_thresh, topk)."""
    rng = np.random.default_rng(seed)
    thr, nms, topk = 0.05, 0.5, 100
    if case == "maskrcnn":  # 1,000 proposals / image, 80 classes, class-specific boxes: a few thousand candidates
        rows, K, Kb, shapes = [1000, 1000], 80, 80, [(800, 1333), (800, 1216)]
    elif case == "agnostic":  # class-agnostic regression: one box per row
        rows, K, Kb, shapes = [700, 300, 512], 20, 1, [(600, 800)] * 3
    elif case == "ragged":  # an empty image, a single row, rows that are not a multiple of anything
        rows, K, Kb, shapes = [0, 1, 37, 2049], 5, 5, [(480, 640)] * 4
    elif case == "nonfinite":  # rows with inf / NaN in a box or a score are dropped as a whole
        rows, K, Kb, shapes = [400, 300], 12, 12, [(512, 512)] * 2
    elif case == "all_pass":  # threshold below every score, no top-k cut
        rows, K, Kb, shapes, thr, topk = [60, 45], 6, 6, [(300, 400)] * 2, -1.0, -1
    elif case == "none_pass":
        rows, K, Kb, shapes, thr = [200, 100], 8, 8, [(300, 400)] * 2, 2.0
    elif case == "many":  # ~25,000 candidates per image: beyond the batched NMS pipeline's 12,288 (per-image launches)
        rows, K, Kb, shapes, thr = [1000, 900], 80, 80, [(800, 1333)] * 2, 0.004
    elif case == "ties":  # quantised scores: many equal scores at the top-k cut and inside the NMS order
        rows, K, Kb, shapes, topk = [600, 500], 10, 10, [(400, 600)] * 2, 50
    else:
        raise ValueError(case)
    boxes, scores = [], []
    for i, r in enumerate(rows):
        h, w = shapes[i]
        ctr = rng.uniform([-20, -20], [w + 20, h + 20], (r, 1, 2))
        if Kb > 1:
            ctr = ctr + rng.normal(0, 6, (r, Kb, 2))
        wh = np.exp(rng.uniform(np.log(8), np.log(300), (r, Kb, 2)))
        b = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=2).reshape(r, Kb * 4).astype(np.float32)
        logits = rng.normal(0, 2.0, (r, K + 1)).astype(np.float32)
        logits[:, -1] += 3.0  # background dominates: a few classes per row pass 0.05
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        s = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        if case == "ties":
            s = (np.round(s * 16) / 16).astype(np.float32)
        if case == "nonfinite" and r:
            bad = rng.choice(r, 25, replace=False)
            b[bad[:8], rng.integers(0, Kb * 4, 8)] = np.inf
            b[bad[8:14], rng.integers(0, Kb * 4, 6)] = np.nan
            s[bad[14:20], rng.integers(0, K + 1, 6)] = np.nan
            s[bad[20:], -1] = -np.inf
        boxes.append(b)
        scores.append(s)
    return boxes, scores, shapes, thr, nms, topk


CASES = ["maskrcnn", "agnostic", "ragged", "nonfinite", "all_pass", "none_pass", "ties", "many"]
