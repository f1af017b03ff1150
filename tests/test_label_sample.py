"""oracle/sampling.py (ROIHeads.label_and_sample_proposals restated for a fixed output size) against
tests/golden/label_sample.npz -- the reference's own pairwise_iou, Matcher and subsample_labels run by
tests/golden/make_golden.py -- and against the sampling contract of modeling/sampling.py:9-54.  CPU only."""
import os

import numpy as np
import pytest

from oracle import sampling as osp

CASES = ["typical", "no_gt", "few", "many_positives", "ignore_band"]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "label_sample.npz"))


def _cfg(g, name):
    c = g[f"{name}_cfg"]
    t = int(c[0])
    return [float(v) for v in c[1:1 + t]], [int(v) for v in c[1 + t:2 + 2 * t]], int(c[-2]), float(c[-1])


def check_sample(out, cand_classes, n_cand, S, num_pos, num_neg, num_classes=80):
    """The contract of subsample_labels, for a fixed-size result: right sizes, positives first, members of their group,
    no duplicates, padding after the last row."""
    cnt = out["counts"]
    assert (int(cnt[0]), int(cnt[1])) == (num_pos, num_pos + num_neg)
    idx = out["index"]
    rows = num_pos + num_neg
    assert len(set(idx[:rows].tolist())) == rows and (idx[:rows] >= 0).all() and (idx[:rows] < n_cand).all()
    cls = cand_classes[idx[:rows]]
    assert np.array_equal(out["classes"][:rows], cls)
    assert ((cls[:num_pos] != -1) & (cls[:num_pos] != num_classes)).all()
    assert (cls[num_pos:] == num_classes).all()
    assert (idx[rows:] == -1).all() and (out["classes"][rows:] == -1).all() and (out["boxes"][rows:] == 0).all()
    assert out["boxes"].shape == (S, 4)


@pytest.mark.parametrize("name", CASES)
def test_labelling_equals_the_reference(golden, name):
    thr, lab, S, frac = _cfg(golden, name)
    cand, idx, mlab, cls = osp.label_candidates(golden[f"{name}_proposals"], golden[f"{name}_gt"],
                                                golden[f"{name}_gt_classes"], thr, lab, 80)
    assert np.array_equal(idx, golden[f"{name}_matched_idxs"])
    assert np.array_equal(mlab, golden[f"{name}_matched_labels"])
    assert np.array_equal(cls, golden[f"{name}_classes"])
    # the reference's draw has the restated sizes and is a valid sample under the checker used for ours
    num_pos, num_neg = osp.sample_sizes(cls, S, frac, 80)
    rp, rn = golden[f"{name}_ref_pos"], golden[f"{name}_ref_neg"]
    assert (len(rp), len(rn)) == (num_pos, num_neg)
    ref_out = {"counts": np.array([num_pos, num_pos + num_neg]), "index": np.concatenate([rp, rn, -np.ones(S - len(rp) - len(rn), np.int64)]),
               "classes": np.concatenate([cls[rp], cls[rn], -np.ones(S - len(rp) - len(rn), np.int64)]),
               "boxes": np.concatenate([cand[rp], cand[rn], np.zeros((S - len(rp) - len(rn), 4), np.float32)])}
    check_sample(ref_out, cls, len(cand), S, num_pos, num_neg)


@pytest.mark.parametrize("name", CASES)
def test_fixed_size_sample(golden, name):
    thr, lab, S, frac = _cfg(golden, name)
    p, g, gc = golden[f"{name}_proposals"], golden[f"{name}_gt"], golden[f"{name}_gt_classes"]
    rng = np.random.default_rng(5)
    keys = rng.random(len(p) + len(g), dtype=np.float32)
    keys[:40] = keys[40:80] if len(p) >= 80 else keys[:40]  # tied keys: broken by candidate index
    out = osp.label_and_sample_fixed(p, len(p), g, gc, keys, thr, lab, S, frac, 80)
    cand, idx, _l, cls = osp.label_candidates(p, g, gc, thr, lab, 80)
    num_pos, num_neg = osp.sample_sizes(cls, S, frac, 80)
    check_sample(out, cls, len(cand), S, num_pos, num_neg)
    rows = num_pos + num_neg
    assert np.array_equal(out["gt_index"][:rows], idx[out["index"][:rows]])
    assert np.array_equal(out["boxes"][:rows], cand[out["index"][:rows]])
    # the smallest keys of each group, in ascending (key, index) order
    ck = np.concatenate([keys[:len(p)], keys[len(p):]])
    for lo, hi, member in ((0, num_pos, (cls != -1) & (cls != 80)), (num_pos, rows, cls == 80)):
        sel = out["index"][lo:hi]
        pairs = [(float(ck[c]), int(c)) for c in sel]
        assert pairs == sorted(pairs)
        rest = [(float(ck[c]), int(c)) for c in np.nonzero(member)[0] if c not in set(sel.tolist())]
        assert not pairs or not rest or max(pairs) < min(rest)


def test_only_the_valid_proposals_count(golden):
    """n_valid < max_p: rows beyond it are not candidates; the appended ground truth keeps its own keys."""
    name = "typical"
    thr, lab, S, frac = _cfg(golden, name)
    p, g, gc = golden[f"{name}_proposals"], golden[f"{name}_gt"], golden[f"{name}_gt_classes"]
    keys = np.random.default_rng(9).random(len(p) + len(g), dtype=np.float32)
    n = 300
    out = osp.label_and_sample_fixed(p, n, g, gc, keys, thr, lab, S, frac, 80)
    short = osp.label_and_sample_fixed(p[:n], n, g, gc, np.concatenate([keys[:n], keys[len(p):]]), thr, lab, S, frac, 80)
    for k in out:
        assert np.array_equal(out[k], short[k]), k
    assert (out["index"] < n + len(g)).all()


def test_masked_mask_loss_oracle_is_the_loss_of_the_foreground_subset():
    """oracle.mask_head.mask_rcnn_loss_masked on a fixed-size list (foreground prefix, background and padding rows) ==
    the pinned mask_rcnn_loss on the foreground rows alone; gradient rows of ignored rows are zero."""
    from oracle import mask_head as omh

    rng = np.random.default_rng(17)
    B, C, M = 24, 80, 28
    x = rng.standard_normal((B, C, M, M)) * 2
    t = rng.random((B, M, M)) < 0.4
    cls = rng.integers(0, C, B)
    cls[9:17] = C   # background rows
    cls[17:] = -1   # padding
    loss, stats, rows = omh.mask_rcnn_loss_masked(x, cls, t)
    want, wstats = omh.mask_rcnn_loss(x[:9], cls[:9], t[:9])
    assert rows == 9 and loss == want and np.array_equal(stats["counts"], wstats["counts"])
    g = omh.mask_rcnn_loss_masked_grad(x, cls, t, 0.5)
    assert np.array_equal(g[:9], omh.mask_rcnn_loss_grad(x[:9], cls[:9], t[:9], 0.5)) and not g[9:].any()
    assert omh.mask_rcnn_loss_masked(x, np.full(B, C), t)[0] == 0.0
