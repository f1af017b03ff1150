"""`subsample_labels` -- detectron2/modeling/sampling.py:9-54, the last step of anchor / proposal labelling
(rpn.py:307-364, roi_heads.py:257-295).  Same contract: (pos_idx, neg_idx), at most
`int(num_samples * positive_fraction)` random positives, the rest filled with random negatives, fewer if there are not
enough; indices are int64 on the labels' device.

The result is RNG-defined (the reference draws two `torch.randperm`s on the device), so there is no bit-level parity to
hold -- what must hold is the distribution: every subset of the right size equally likely.  This version draws ONE
uniform key per element and takes the smallest keys of each class (a uniform random subset), with ONE host sync (the
two class counts) instead of the reference's two `nonzero` syncs and two device sorts inside `randperm`.  Plain
PyTorch on purpose: there is no arithmetic here to put on the matrix cores, and the random stream is torch's
(`generator` / the global seed), so runs reproduce under `torch.manual_seed` like the reference's."""
import torch

__all__ = ["subsample_labels"]


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int,
                     generator: torch.Generator = None):
    """labels (N,): -1 ignore, bg_label negative, anything else positive.  Returns (pos_idx, neg_idx)."""
    assert labels.dim() == 1, labels.shape
    n = labels.numel()
    dev = labels.device
    if n == 0:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        return e, e.clone()
    pos_mask = (labels != -1) & (labels != bg_label)
    neg_mask = labels == bg_label
    counts = torch.stack([pos_mask.sum(), neg_mask.sum()]).tolist()  # the one host sync
    num_pos = min(counts[0], int(num_samples * positive_fraction))
    num_neg = min(counts[1], num_samples - num_pos)
    key = torch.rand(n, device=dev, generator=generator)
    two = torch.full((), 2.0, device=dev)  # sorts after every real key
    pos_idx = torch.topk(torch.where(pos_mask, key, two), num_pos, largest=False, sorted=False).indices
    neg_idx = torch.topk(torch.where(neg_mask, key, two), num_neg, largest=False, sorted=False).indices
    return pos_idx, neg_idx
