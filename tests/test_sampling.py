"""subsample_labels (detectron2/modeling/sampling.py:9-54): RNG-defined, so the checks are the contract -- sizes,
membership, the fill rule -- and the distribution (every element of a class equally likely), on CPU tensors (plain
PyTorch; the GPU suite repeats the contract on the device)."""
import pytest
import torch

from detectron2_amd.modeling import subsample_labels


def _labels(n_pos, n_neg, n_ign, bg=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    lab = torch.cat([torch.randint(1, 80, (n_pos,), generator=g), torch.full((n_neg,), bg), torch.full((n_ign,), -1)])
    return lab[torch.randperm(lab.numel(), generator=g)]


@pytest.mark.parametrize("n_pos,n_neg,num,frac,exp", [
    (300, 5000, 512, 0.25, (128, 384)),   # plenty of both: int(num * frac) positives, the rest negatives
    (20, 5000, 512, 0.25, (20, 492)),     # few positives: negatives fill the sample
    (300, 100, 512, 0.25, (128, 100)),    # few negatives: fewer than num_samples in total
    (0, 50, 64, 0.5, (0, 50)), (10, 0, 64, 0.5, (10, 0)), (0, 0, 64, 0.5, (0, 0)),
])
def test_sizes_membership_and_fill_rule(n_pos, n_neg, num, frac, exp):
    lab = _labels(n_pos, n_neg, 40)
    pos, neg = subsample_labels(lab, num, frac, 0)
    assert (len(pos), len(neg)) == exp and pos.dtype == neg.dtype == torch.int64
    assert len(set(pos.tolist())) == len(pos) and len(set(neg.tolist())) == len(neg)   # no index twice
    assert bool(((lab[pos] != -1) & (lab[pos] != 0)).all()) and bool((lab[neg] == 0).all())


def test_uniform_and_reproducible():
    lab = _labels(40, 200, 10, bg=80)  # bg_label = num_classes, as the ROI heads use it
    hits = torch.zeros(lab.numel())
    g = torch.Generator().manual_seed(7)
    runs = 3000
    for _ in range(runs):
        pos, neg = subsample_labels(lab, 32, 0.25, 80, generator=g)
        hits[pos] += 1
        hits[neg] += 1
    is_pos, is_neg = (lab != -1) & (lab != 80), lab == 80
    # each of the 40 positives is drawn with probability 8 / 40, each of the 200 negatives with 24 / 200
    assert (hits[is_pos] / runs - 8 / 40).abs().max() < 0.04 and (hits[is_neg] / runs - 24 / 200).abs().max() < 0.03
    assert hits[lab == -1].sum() == 0
    a = subsample_labels(lab, 32, 0.25, 80, generator=torch.Generator().manual_seed(3))
    b = subsample_labels(lab, 32, 0.25, 80, generator=torch.Generator().manual_seed(3))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    e = subsample_labels(torch.zeros(0, dtype=torch.int64), 16, 0.5, 0)
    assert len(e[0]) == len(e[1]) == 0
