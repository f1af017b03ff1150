"""subsample_labels (detectron2/modeling/sampling.py:9-54).  The reference's result is RNG-defined, so what is checked
on the CPU is the ORACLE's restatement with explicit keys (oracle/sampling.py: subsample_labels_keys, the rule
d2amd_subsample_labels implements): the contract -- sizes, membership, the fill rule -- against the reference's own
function where /root/reference exists, and the distribution (every element of a group equally likely).  The device
kernel is held to the oracle bit for bit in tests/test_gpu_subsample.py; the product has no CPU path."""
import numpy as np
import pytest
import torch

from oracle import ref
from oracle import sampling as osp


def _labels(n_pos, n_neg, n_ign, bg=0, seed=0):
    rng = np.random.default_rng(seed)
    lab = np.concatenate([rng.integers(1, 80, n_pos), np.full(n_neg, bg), np.full(n_ign, -1)]).astype(np.int64)
    return lab[rng.permutation(lab.size)]


CASES = [
    (300, 5000, 512, 0.25, (128, 384)),   # plenty of both: int(num * frac) positives, the rest negatives
    (20, 5000, 512, 0.25, (20, 492)),     # few positives: negatives fill the sample
    (300, 100, 512, 0.25, (128, 100)),    # few negatives: fewer than num_samples in total
    (0, 50, 64, 0.5, (0, 50)), (10, 0, 64, 0.5, (10, 0)), (0, 0, 64, 0.5, (0, 0)),
]


@pytest.mark.parametrize("n_pos,n_neg,num,frac,exp", CASES)
def test_sizes_membership_and_fill_rule(n_pos, n_neg, num, frac, exp):
    lab = _labels(n_pos, n_neg, 40)
    keys = np.random.default_rng(1).random(lab.size, dtype=np.float32)
    pos, neg = osp.subsample_labels_keys(lab, keys, num, frac, 0)
    assert (len(pos), len(neg)) == exp and pos.dtype == neg.dtype == np.int64
    assert len(set(pos.tolist())) == len(pos) and len(set(neg.tolist())) == len(neg)   # no index twice
    assert ((lab[pos] != -1) & (lab[pos] != 0)).all() and (lab[neg] == 0).all()
    # the smallest keys of each group, in ascending order
    for idx, member in ((pos, (lab != -1) & (lab != 0)), (neg, lab == 0)):
        if len(idx):
            assert (np.diff(keys[idx]) >= 0).all()
            rest = np.setdiff1d(np.nonzero(member)[0], idx)
            assert len(rest) == 0 or keys[rest].min() >= keys[idx].max()


@pytest.mark.parametrize("n_pos,n_neg,num,frac,exp", CASES)
def test_same_sizes_and_groups_as_the_reference_function(n_pos, n_neg, num, frac, exp):
    """The reference's own subsample_labels on the same labels: same result sizes, members of the same groups."""
    from conftest import need_reference

    need_reference(ref.have_py(), "the reference's modeling/sampling.py (oracle/_ref/py)")
    rs = ref.py_sampling()
    lab = _labels(n_pos, n_neg, 40, seed=3)
    rp, rn = rs.subsample_labels(torch.from_numpy(lab), num, frac, 0)
    pos, neg = osp.subsample_labels_keys(lab, np.random.default_rng(2).random(lab.size, dtype=np.float32), num, frac, 0)
    assert (len(rp), len(rn)) == (len(pos), len(neg)) == exp
    assert set(np.unique(lab[rp.numpy()]) if len(rp) else []) <= set(lab[(lab != -1) & (lab != 0)])
    assert (lab[rn.numpy()] == 0).all() and (lab[neg] == 0).all()


def test_ties_go_to_the_lower_index_and_zero_signs_tie():
    lab = np.array([1, 1, 0, 0, 1, 0, -1, 0], np.int64)
    keys = np.array([0.5, 0.5, 0.25, -0.0, 0.5, 0.0, 0.0, 0.25], np.float32)
    pos, neg = osp.subsample_labels_keys(lab, keys, 5, 0.4, 0)
    assert pos.tolist() == [0, 1] and neg.tolist() == [3, 5, 2]


def test_uniform():
    lab = _labels(40, 200, 10, bg=80)  # bg_label = num_classes, as the ROI heads use it
    hits = np.zeros(lab.size)
    rng = np.random.default_rng(7)
    runs = 3000
    for _ in range(runs):
        pos, neg = osp.subsample_labels_keys(lab, rng.random(lab.size, dtype=np.float32), 32, 0.25, 80)
        hits[pos] += 1
        hits[neg] += 1
    is_pos, is_neg = (lab != -1) & (lab != 80), lab == 80
    # each of the 40 positives is drawn with probability 8 / 40, each of the 200 negatives with 24 / 200
    assert np.abs(hits[is_pos] / runs - 8 / 40).max() < 0.04 and np.abs(hits[is_neg] / runs - 24 / 200).max() < 0.03
    assert hits[lab == -1].sum() == 0


def test_anchor_label_rewrite():
    lab = _labels(30, 900, 70).clip(-1, 1).astype(np.int8)
    keys = np.random.default_rng(5).random(lab.size, dtype=np.float32)
    out = osp.subsample_anchor_labels(lab, keys, 256, 0.5)
    assert out.dtype == np.int8 and (out == 1).sum() == 30 and (out == 0).sum() == 226 and (out == -1).sum() == 744
    assert (lab[out == 1] == 1).all() and (lab[out == 0] == 0).all()


def test_product_has_no_cpu_path():
    from detectron2_amd.modeling import subsample_labels

    with pytest.raises(NotImplementedError):
        subsample_labels(torch.zeros(8, dtype=torch.int64), 4, 0.5, 0)


def test_philox_restatement_known_answers():
    """oracle.sampling.philox_uniform_keys (the restatement of csrc/random_keys.hip) against the published known-answer
    vectors of Philox4x32-10 (Random123 kat_vectors: counter / key all zero, all ones, digits of pi)."""
    k = osp.philox_uniform_keys(0, 0, 4)
    assert [int(x) for x in (k.astype(np.float64) * 2 ** 24)] == [0x6627e8d5 >> 8, 0xe169c58d >> 8, 0xbc57ac4c >> 8, 0x9b00dbd8 >> 8]
    # thread q = 0xffffffff_ffffffff cannot be reached through the array interface; offset / seed all ones, thread 0:
    k = osp.philox_uniform_keys((1 << 64) - 1, (1 << 64) - 1, 4)
    assert k.dtype == np.float32 and (k >= 0).all() and (k < 1).all()
    # a long draw: uniform to 3 sigma in every decile, no value repeated suspiciously often
    k = osp.philox_uniform_keys(1234, 7, 400000)
    h = np.histogram(k, 10, (0, 1))[0]
    assert np.abs(h - 40000).max() < 3.5 * np.sqrt(40000 * 0.9)
