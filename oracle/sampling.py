"""CPU restatement of ROIHeads.label_and_sample_proposals -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Follows detectron2/modeling/roi_heads/roi_heads.py:219-295 step by step:
  add_ground_truth_to_proposals (proposal_generator/proposal_utils.py:138-205: ground-truth boxes appended AFTER the
  proposals), pairwise_iou + Matcher (oracle.pairwise_iou / oracle.matcher: the C restatements pinned to
  tests/golden/pairwise_iou.npz and matcher.npz), the relabelling of _sample_proposals (roi_heads.py:199-208) and
  subsample_labels (modeling/sampling.py:9-54).

The deterministic part (matched index, matcher label, class per candidate, the two sample sizes) is pinned to the
reference's own functions by tests/golden/label_sample.npz (tests/golden/make_golden.py).  The random part is
RNG-defined in the reference (two torch.randperm draws); the restated rule is the one of
detectron2_amd/modeling/sampling.py -- a uniform key per candidate, the smallest keys of each group, ties by index --
written for a FIXED output size: positives, negatives, padding (see include/d2amd.h:
d2amd_label_and_sample_proposals).
"""
import numpy as np

import oracle


def label_candidates(proposals, gt_boxes, gt_classes, thresholds, labels, num_classes, append_gt=True):
    """-> (candidate boxes [n + G, 4], matched_idxs [n + G] int64, matched_labels int8, classes int64)."""
    proposals = np.asarray(proposals, np.float32).reshape(-1, 4)
    gt_boxes = np.asarray(gt_boxes, np.float32).reshape(-1, 4)
    gt_classes = np.asarray(gt_classes, np.int64).reshape(-1)
    cand = np.concatenate([proposals, gt_boxes], 0) if append_gt else proposals  # proposal_utils.py:196-203
    if len(gt_boxes) == 0:  # matcher.py:80-90 + roi_heads.py:207
        idx = np.zeros(len(cand), np.int64)
        lab = np.full(len(cand), labels[0], np.int8)
        return cand, idx, lab, np.full(len(cand), num_classes, np.int64)
    q = oracle.pairwise_iou(gt_boxes, cand)  # roi_heads.py:267-269
    idx, lab = oracle.matcher(q, thresholds, labels, False)  # roi_heads.py:270
    cls = gt_classes[idx].copy()  # roi_heads.py:201
    cls[lab == 0] = num_classes  # :203
    cls[lab == -1] = -1  # :205
    return cand, idx, lab, cls


def sample_sizes(classes, num_samples, positive_fraction, bg_label):
    """sampling.py:39-47 -> (num_pos, num_neg)."""
    pos = int(((classes != -1) & (classes != bg_label)).sum())
    neg = int((classes == bg_label).sum())
    num_pos = min(pos, int(num_samples * positive_fraction))
    return num_pos, min(neg, num_samples - num_pos)


def subsample_labels_keys(labels, keys, num_samples, positive_fraction, bg_label):
    """subsample_labels (sampling.py:9-54) with the randomness made explicit: `keys` = one uniform number per element;
    the num_pos smallest keys among the positives and the num_neg smallest among the negatives (ties towards the lower
    index), each list in ascending (key, index) order -- the rule of csrc/subsample.hip (d2amd_subsample_labels).
    -> (pos_idx int64, neg_idx int64)."""
    labels = np.asarray(labels).reshape(-1).astype(np.int64)
    keys = np.asarray(keys, np.float32).reshape(-1)
    assert labels.shape == keys.shape
    num_pos, num_neg = sample_sizes(labels, num_samples, positive_fraction, bg_label)
    order = np.lexsort((np.arange(len(labels)), keys))
    is_pos = ((labels != -1) & (labels != bg_label))[order]
    is_neg = (labels == bg_label)[order]
    return order[is_pos][:num_pos].astype(np.int64), order[is_neg][:num_neg].astype(np.int64)


def subsample_anchor_labels(labels, keys, num_samples, positive_fraction):
    """RPN._subsample_labels (proposal_generator/rpn.py:287-305) on one label vector, keys explicit."""
    pos, neg = subsample_labels_keys(labels, keys, num_samples, positive_fraction, 0)
    out = np.full(np.asarray(labels).reshape(-1).shape, -1, np.int8)  # rpn.py:300
    out[pos] = 1  # :301
    out[neg] = 0  # :302
    return out


def label_and_sample_fixed(proposals, n_valid, gt_boxes, gt_classes, keys, thresholds=(0.5,), labels=(0, 1),
                           batch_size_per_image=512, positive_fraction=0.25, num_classes=80, append_gt=True):
    """proposals [max_p, 4] of which the first n_valid count; keys [max_p + G].  -> dict of fixed-size arrays:
    boxes [S, 4], classes [S], gt_index [S], index [S], counts (num_pos, rows)."""
    proposals = np.asarray(proposals, np.float32).reshape(-1, 4)
    max_p = proposals.shape[0]
    n = max(0, min(int(n_valid), max_p))
    keys = np.asarray(keys, np.float32).reshape(-1)
    G = np.asarray(gt_boxes).reshape(-1, 4).shape[0]
    cand, idx, _lab, cls = label_candidates(proposals[:n], gt_boxes, gt_classes, thresholds, labels, num_classes,
                                            append_gt)
    ckeys = np.concatenate([keys[:n], keys[max_p:max_p + G]]) if append_gt else keys[:n]
    S = batch_size_per_image
    num_pos, num_neg = sample_sizes(cls, S, positive_fraction, num_classes)
    order = np.lexsort((np.arange(len(cand)), ckeys))  # ascending key, ties by candidate index
    is_pos = (cls != -1) & (cls != num_classes)
    is_neg = cls == num_classes
    pos = [c for c in order if is_pos[c]][:num_pos]
    neg = [c for c in order if is_neg[c]][:num_neg]
    sel = np.asarray(pos + neg, np.int64)
    out = {
        "boxes": np.zeros((S, 4), np.float32),
        "classes": np.full(S, -1, np.int64),
        "gt_index": np.zeros(S, np.int64),
        "index": np.full(S, -1, np.int64),
        "counts": np.asarray([num_pos, num_pos + num_neg], np.int32),
    }
    k = len(sel)
    if k:
        out["boxes"][:k] = cand[sel]
        out["classes"][:k] = cls[sel]
        out["gt_index"][:k] = idx[sel]
        out["index"][:k] = sel
    return out


def philox_uniform_keys(seed, offset, n):
    """d2amd_uniform_keys restated (csrc/random_keys.hip): Philox4x32-10 (Salmon et al., SC'11) with counter
    {thread lo, thread hi, offset lo, offset hi}, key = seed; thread q yields outputs 4 q .. 4 q + 3 = (x >> 8) * 2^-24."""
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    c = [q & 0xFFFFFFFF, q >> np.uint64(32), np.full_like(q, offset & 0xFFFFFFFF), np.full_like(q, (offset >> 32) & 0xFFFFFFFF)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    out = np.stack(c, 1).reshape(-1)[:n]
    return ((out >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
